"""API-level timing of the facade (libuhdr.so): uhdr_encode / uhdr_decode of a 4K frame with and without
uhdr_enable_gpu_acceleration, with the seam's stage trace (UHDR_HIP_SEAM_TRACE=1 shows a timestamp per stage)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from libultrahdr_amd import capi as A  # noqa: E402
from libultrahdr_amd import facade as FA  # noqa: E402
from libultrahdr_amd import synth  # noqa: E402


def med(fn, n):
    ts = []
    r = None
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return r, sorted(ts)[len(ts) // 2] * 1e3


def main():
    w, h = 3840, 2160
    multi = "multi" in sys.argv  # 3-channel gain map at full resolution (BASELINE config 2's map C)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
    sdr = synth.make_sdr_yuv420(w, h)
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    out = {}
    FA.encode(hdr, sdr, gpu=True)
    jpg, out["encode_gpu_ms"] = med(lambda: FA.encode(hdr, sdr, gpu=True), 3)
    _, out["decode_gpu_ms"] = med(lambda: FA.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=True), 5)
    if "cpu" in sys.argv:
        _, out["encode_cpu_ms"] = med(lambda: FA.encode(hdr, sdr, gpu=False), 1)
        _, out["decode_cpu_ms"] = med(lambda: FA.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=False), 2)
    out["jpeg_bytes"] = len(jpg)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
