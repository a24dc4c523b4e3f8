"""Hunt for a tone-map sample where the HIP path and the real reference (oracle/_ref) differ: random small P010 / RGBA1010102 frames until one
differs; prints the frame's parameters, the plane / position, both outputs, the port's (oracle C restatement) output and the input sample."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from libultrahdr_amd.ultrahdr import Context, UltraHdr
from oracle import loader as L

ctx = Context(0)
u = UltraHdr(ctx=ctx)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
t0 = time.time()
cases = found = 0
while time.time() - t0 < float(sys.argv[2] if len(sys.argv) > 2 else 120) and found < 3:
    w, h = int(rng.choice([130, 256, 512])), int(rng.choice([66, 128, 256]))
    kind = rng.choice(["p010", "1010102"])
    ct = int(rng.choice([A.UHDR_CT_HLG, A.UHDR_CT_PQ, A.UHDR_CT_LINEAR]))
    cg = int(rng.integers(0, 3))
    seed = int(rng.integers(1 << 30))
    hdr = (synth.make_hdr_p010(w, h, seed=seed, ct=ct, cg=cg, noise=0.06) if kind == "p010" else synth.make_hdr_rgba1010102(w, h, seed=seed, ct=ct, cg=cg, noise=0.06))
    want = L.tone_map("ref", hdr)
    got = Image(want.fmt, w, h, align=64)
    u.toneMap(hdr, got)
    cases += 1
    for pi, (pg, pw) in enumerate(zip(got.planes_valid(), want.planes_valid())):
        if pg.dtype == np.uint32:
            pg, pw = pg.view(np.uint8), pw.view(np.uint8)
        d = pg.astype(np.int32) - pw.astype(np.int32)
        if (d != 0).any():
            found += 1
            port = L.tone_map("port", hdr)
            pp = port.planes_valid()[pi]
            if pp.dtype == np.uint32:
                pp = pp.view(np.uint8)
            ys, xs = np.nonzero(d)
            print(f"case {cases}: {kind} {w}x{h} ct{ct} cg{cg} seed {seed} plane {pi}: {len(ys)} samples differ")
            for y, x in list(zip(ys, xs))[:4]:
                print(f"   (y {y}, x {x}): hip {int(pg[y, x])}  reference {int(pw[y, x])}  oracle restatement {int(pp[y, x])}")
            break
print(f"{cases} frames, {found} with a difference, {time.time() - t0:.0f} s")
