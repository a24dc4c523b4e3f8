#!/usr/bin/env python
"""A/B of the round-4 review's proposal for the HLG / PQ decode tail (config 5): 4-byte output-code bucket entries (threshold
offset | code, one ds_read_b32 + four VALU instructions per channel) against the product's 8-byte entries (threshold, lo | hi:
one ds_read_b64 + three).  The 4-byte form is apply_gainmap.hip compiled with -DUHDR_EXP_CODE4 and linked with the product's
other objects into tools/_exp/libuhdr_hip_code4.so (an experiment, not shipped).  Run once per library on the same box:
    python tools/code4_exp.py                       # product library
    UHDR_EXP_LIB=tools/_exp/libuhdr_hip_code4.so python tools/code4_exp.py
Prints the config-5 figures (batch of 32 4K frames -> HLG, HIP-graph replay and eager HIP-event time), the same batch to PQ, and
SHA-256 digests of all output frames: the two libraries must agree on every digest."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libultrahdr_amd import capi as A  # noqa: E402

if os.environ.get("UHDR_EXP_LIB"):
    A.LIB_PATH = os.path.join(ROOT, os.environ["UHDR_EXP_LIB"])
import torch  # noqa: E402

import bench  # noqa: E402
from libultrahdr_amd import synth  # noqa: E402
from libultrahdr_amd.ultrahdr import Context, UltraHdr  # noqa: E402

torch.cuda.set_device(0)
ctx = Context(0)
u = UltraHdr(ctx=ctx)
out = {"library": os.path.relpath(A.LIB_PATH, ROOT)}
c5 = bench.config5_section(ctx, u, "cuda:0")
out["hlg_graph_replay_us"] = c5["graph_replay_us_per_batch"]
out["hlg_eager_kernel_us"] = c5["kernel_us_per_batch_eager_hip_events"]
out["hlg_frac"] = c5["frac_of_8TBs"]

nb, w, h = 32, 3840, 2160
u32 = A.UHDR_IMG_FMT_32bppRGBA1010102
md = synth.default_metadata(use_base_cg=0)
sets = bench.make_frames(nb, w, h, "A", "cuda:0", u32, seed0=555)
for s5, g5, _ in sets:
    s5.raw.cg, g5.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
for name, ct in (("hlg", A.UHDR_CT_HLG), ("pq", A.UHDR_CT_PQ)):
    args = ([f[0] for f in sets], [f[1] for f in sets], md, ct, u32, A.FLT_MAX, [f[2] for f in sets])
    u.applyGainMapBatch(*args)
    ctx.synchronize()
    hs = hashlib.sha256()
    for f in sets:
        hs.update(f[2].buf.cpu().numpy().tobytes())
    out[f"{name}_digest"] = hs.hexdigest()[:16]
    bench.clock_ramp(ctx, lambda: u.applyGainMapBatch(*args))
    ms = bench.time_kernel(ctx, lambda: u.applyGainMapBatch(*args), iters=20, warm=3)
    out[f"{name}_eager_kernel_us_after_ramp"] = round(ms * 1e3, 1)
# three-channel maps (the variants that run at three waves per SIMD), one 4K frame each
from libultrahdr_amd.images import Image  # noqa: E402

for name, ct in (("hlg", A.UHDR_CT_HLG), ("pq", A.UHDR_CT_PQ)):
    sdr = synth.make_sdr_yuv420(w, h).to("cuda:0")
    gm = synth.make_gainmap(w // 2, h // 2, 3, True, seed=9, cg=A.UHDR_CG_BT_2100).to("cuda:0")
    dst = Image(u32, w, h, align=64, device="cuda:0")
    fn = lambda: u.applyGainMap(sdr, gm, md, ct, u32, A.FLT_MAX, dst)  # noqa: E731
    fn()
    ctx.synchronize()
    out[f"{name}_map3ch_s2_digest"] = hashlib.sha256(dst.buf.cpu().numpy().tobytes()).hexdigest()[:16]
    bench.clock_ramp(ctx, fn, seconds=0.5)
    out[f"{name}_map3ch_s2_us"] = round(bench.time_kernel(ctx, fn, iters=30, warm=5) * 1e3, 2)
print(json.dumps(out))
