#!/usr/bin/env python
"""Counts the samples on which the HIP encode kernels differ from the oracle (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from libultrahdr_amd.ultrahdr import Context, UltraHdr
from oracle import loader as L
import test_gpu_parity as T

ctx = Context(0)
w, h = 1280, 720
kind = "ref" if os.path.exists(os.path.join(os.path.dirname(L.__file__), "_ref", "libuhdr_ref.so")) else "port"
print("oracle:", kind)
for hdr_kw, sdr_kind, cfg_kw in T.GEN_CASES:
    sdr, hdr = T._pair(w, h, hdr_kw, sdr_kind)
    cfg = A.default_encode_cfg(**cfg_kw)
    md_w, gm_w = L.generate_gainmap(kind, sdr, hdr, cfg)
    md_g, gm_g = T._uhdr_for(ctx, cfg).generateGainMap(sdr, hdr, bool(cfg.sdr_is_601), bool(cfg.use_luminance))
    d = np.abs(gm_g.valid(0).astype(np.int32) - gm_w.valid(0).astype(np.int32))
    print("generate", hdr_kw, sdr_kind, cfg_kw, "differ:", int((d != 0).sum()), "of", d.size, "max", int(d.max()),
          "md equal:", md_g.as_dict() == md_w.as_dict())
u = UltraHdr(ctx=ctx)
for k, ct, cg in [("p010", A.UHDR_CT_HLG, A.UHDR_CG_BT_2100), ("p010", A.UHDR_CT_PQ, A.UHDR_CG_DISPLAY_P3),
                  ("1010102", A.UHDR_CT_PQ, A.UHDR_CG_BT_2100), ("1010102", A.UHDR_CT_HLG, A.UHDR_CG_BT_709),
                  ("p010", A.UHDR_CT_LINEAR, A.UHDR_CG_BT_2100)]:
    hdr = synth.make_hdr_p010(w, h, ct=ct, cg=cg) if k == "p010" else synth.make_hdr_rgba1010102(w, h, ct=ct, cg=cg)
    want = L.tone_map(kind, hdr)
    got = Image(want.fmt, w, h, align=64)
    u.toneMap(hdr, got)
    n = tot = mx = 0
    for pg, pw in zip(got.planes_valid(), want.planes_valid()):
        if pg.dtype == np.uint32:
            pg, pw = pg.view(np.uint8), pw.view(np.uint8)
        d = np.abs(pg.astype(np.int32) - pw.astype(np.int32))
        n += int((d != 0).sum()); tot += d.size; mx = max(mx, int(d.max()))
    print("tonemap", k, ct, cg, "differ:", n, "of", tot, "max", mx)
