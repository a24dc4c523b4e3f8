#!/usr/bin/env python
"""Where the time of one accelerated uhdr_encode / uhdr_decode goes: the facade's stage trace (UHDR_HIP_SEAM_TRACE: one line
per stage with a timestamp and its duration, plus the begin / end of the call) for the 4K API-1 encode and the 4K F16
decode of bench.py's api_level section.  The gaps between the stage lines are the reference's own host code (container
parsing, buffer allocation, copy_raw_image, ...).  Run on the GPU box: python tools/trace_api.py 2> trace.txt"""
import os
import sys

os.environ["UHDR_HIP_SEAM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libultrahdr_amd import capi as A  # noqa: E402
from libultrahdr_amd import facade as FA  # noqa: E402
from libultrahdr_amd import synth  # noqa: E402

w, h = 3840, 2160
hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
sdr = synth.make_sdr_yuv420(w, h)
for i in range(8):
    print(f"--- uhdr_encode #{i}", file=sys.stderr, flush=True)
    jpg = FA.encode(hdr, sdr, gpu=True)
    print(f"--- uhdr_encode took {FA.last_call_seconds * 1e3:.2f} ms", file=sys.stderr, flush=True)
for i in range(4):
    print(f"--- uhdr_decode #{i}", file=sys.stderr, flush=True)
    FA.decode(jpg, A.UHDR_CT_LINEAR, A.UHDR_IMG_FMT_64bppRGBAHalfFloat, gpu=True)
    print(f"--- uhdr_decode took {FA.last_call_seconds * 1e3:.2f} ms", file=sys.stderr, flush=True)
print("--- uhdr_decode + uhdr_get_decoded_gainmap_image", file=sys.stderr, flush=True)
FA.decode(jpg, A.UHDR_CT_LINEAR, A.UHDR_IMG_FMT_64bppRGBAHalfFloat, gpu=True, want_gainmap=True)
print(f"--- uhdr_decode took {FA.last_call_seconds * 1e3:.2f} ms (the gain-map image download comes after it)", file=sys.stderr, flush=True)
