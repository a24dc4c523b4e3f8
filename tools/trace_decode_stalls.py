#!/usr/bin/env python
"""How often, and where, a 4K uhdr_decode through the facade stalls: N decodes back to back (the harness copies the 66 MB result
between them), per call the C call's time and the seam's stage trace.  Run under different switches to attribute the stalls:
    python tools/trace_decode_stalls.py [n]          UHDR_HIP_UPLOAD_THREADS=0 python tools/trace_decode_stalls.py"""
import os
import sys

os.environ["UHDR_HIP_SEAM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libultrahdr_amd import capi as A  # noqa: E402
from libultrahdr_amd import facade as FA  # noqa: E402
from libultrahdr_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w, h = 3840, 2160
hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
sdr = synth.make_sdr_yuv420(w, h)
jpg = FA.encode(hdr, sdr, gpu=True)
jpg = FA.encode(hdr, sdr, gpu=True)
ts = []
for i in range(n):
    print(f"--- uhdr_decode #{i}", file=sys.stderr, flush=True)
    FA.decode(jpg, A.UHDR_CT_LINEAR, A.UHDR_IMG_FMT_64bppRGBAHalfFloat, gpu=True)
    ts.append(FA.last_call_seconds * 1e3)
print("uhdr_decode ms:", " ".join(f"{t:.1f}" for t in ts), file=sys.stderr)
print("sorted:", " ".join(f"{t:.1f}" for t in sorted(ts)), file=sys.stderr)
