import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from libultrahdr_amd import capi as A, synth
from libultrahdr_amd.images import Image
from libultrahdr_amd.ultrahdr import Context, UltraHdr
from oracle import loader as L
ctx = Context(0); u = UltraHdr(ctx=ctx)
f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
for (w, h) in [(256, 128), (512, 64), (1024, 256)]:
  for mk in "ABC":
    for ubc in (0, 1):
      sdr = synth.make_sdr_yuv420(w, h, seed=3)
      gm = synth.make_gainmap(w // 4, h // 4, 1, seed=5) if mk == "A" else synth.make_gainmap(w, h, 3, alpha=(mk == "C"), seed=5)
      sdr.raw.cg, gm.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
      md = synth.default_metadata(use_base_cg=ubc)
      for ct, fmt in ((A.UHDR_CT_LINEAR, f16), (A.UHDR_CT_HLG, A.UHDR_IMG_FMT_32bppRGBA1010102)):
        dest = Image(fmt, w, h, align=2, device="cuda:0")
        u.applyGainMap(sdr.to("cuda:0"), gm.to("cuda:0"), md, ct, fmt, A.FLT_MAX, dest)
        ctx.synchronize()
        want = L.apply_gainmap("port", sdr, gm, md, ct).valid(0)
        got = dest.to_host().valid(0)
        bad = np.argwhere(got != want)
        msg = "ok" if len(bad) == 0 else f"{len(bad)} of {got.size} differ; first {bad[:3].tolist()} got {[hex(int(got[tuple(b)])) for b in bad[:3]]} want {[hex(int(want[tuple(b)])) for b in bad[:3]]} rows {sorted(set(bad[:,0].tolist()))[:8]} cols min {bad[:,1].min()} max {bad[:,1].max()}"
        print(w, h, mk, "ubc", ubc, "ct", ct, msg, flush=True)
