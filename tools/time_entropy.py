#!/usr/bin/env python
"""Times the entropy stage (Huffman encode / decode) of a 4K 4:2:0 q95 frame for several restart intervals (HIP events
around the library's launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

from libultrahdr_amd import synth
from libultrahdr_amd.ultrahdr import Context, UltraHdr

w, h = 3840, 2160
ctx = Context(0)
u = UltraHdr(ctx=ctx)
sdr = synth.make_sdr_yuv420(w, h).to("cuda:0")
qt = [u.quant_table(95, False), u.quant_table(95, True), u.quant_table(95, True)]
coefs = []
for c in range(3):
    rows, stride, wv = sdr.layout[c]
    coefs.append(u.fdct_quant(sdr.plane_tensor(c), stride, wv // 8, rows // 8, qt[c]))
out = torch.empty(w * h * 2, dtype=torch.uint8, device="cuda:0")
sampling = [(2, 2), (1, 1), (1, 1)]
shapes = [tuple(c.shape[:2]) for c in coefs]


def timed(family, fn, iters=3):
    fn()
    ctx.synchronize()
    ctx.profile(True)
    ctx.profile_read(None, reset=True)
    for _ in range(iters):
        fn()
    n, ms = ctx.profile_read(family, reset=True)
    ctx.profile(False)
    return ms / max(n, 1) * 1e3


for ri in (1, 2, 3, 4, 5, 10):
    stream = u.huffman_encode(coefs, w, h, sampling, ri, out=out).clone()
    t_enc = timed("huffman_encode", lambda: u.huffman_encode(coefs, w, h, sampling, ri, out=out))
    t_dec = timed("huffman_decode", lambda: u.huffman_decode(stream, shapes, w, h, sampling, ri))
    back = u.huffman_decode(stream, shapes, w, h, sampling, ri)
    ok = all(torch.equal(a, b) for a, b in zip(back, coefs))
    print(f"ri={ri:2d} stream {stream.numel()} B  encode {t_enc:7.1f} us  decode {t_dec:7.1f} us  round trip {'ok' if ok else 'MISMATCH'}")
