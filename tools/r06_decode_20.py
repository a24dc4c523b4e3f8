"""20 back-to-back accelerated uhdr_decode calls of one 4K file through the facade: the time of every call (ms).  Round 5 saw one call in
four stall for 10-16 ms in the runtime's pageable upload of the compressed scans and dodged it with a thread-lifetime staging copy in the seam;
round 6 sends the bytes through the library's pinned ring instead (fast_h2d) and has no such copy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libultrahdr_amd import capi as A
from libultrahdr_amd import facade as FA
from libultrahdr_amd import synth

w, h = 3840, 2160
hdr, sdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG), synth.make_sdr_yuv420(w, h)
jpg = FA.encode(hdr, sdr, gpu=True)
f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
FA.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=True)
ts = []
for _ in range(20):
    FA.decode(bytes(jpg), A.UHDR_CT_LINEAR, f16, gpu=True)  # a fresh copy of the file each time, as an application would hold it
    ts.append(FA.last_call_seconds * 1e3)
print("uhdr_decode 4K -> RGBA_F16, 20 calls, ms:", " ".join(f"{t:.2f}" for t in ts))
print(f"max {max(ts):.2f} ms, median {sorted(ts)[10]:.2f} ms")
te = []
for _ in range(10):
    FA.encode(hdr, sdr, gpu=True)
    te.append(FA.last_call_seconds * 1e3)
print("uhdr_encode API-1 4K, 10 calls, ms:", " ".join(f"{t:.2f}" for t in te))
