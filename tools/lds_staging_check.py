#!/usr/bin/env python
"""Which kernels stage global memory into LDS one element at a time?  Compiles every .hip translation unit to gfx950 ISA and counts, per
kernel, the places where a global load is followed -- with nothing but address arithmetic in between -- by a wait for ALL outstanding loads
and an LDS store of the loaded registers: one memory latency per element (round 6: lds_copy.h's first form was compiled to exactly that).

    python tools/lds_staging_check.py            (prints kernels with such places; exit 1 if one of them has seven or more -- an unrolled
                                                  staging loop; one to six are single look-up-table words a thread fetches once)
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "libultrahdr_amd", "csrc")


def isa(path):
    out = "/tmp/_stg_" + os.path.basename(path) + ".s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fwrapv", "-fvisibility=hidden",
                           "-I" + SRC, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", path, "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    worst = 0
    for path in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        text = isa(path)
        for m in re.finditer(r"^(_Z[^\n:]*):\s*;[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", text, re.S | re.M):
            ins = [l.strip() for l in m.group(2).split("\n") if l.startswith("\t") and not l.strip().startswith((";", "."))]
            n = 0
            for i, l in enumerate(ins):
                mm = re.match(r"(?:global|buffer)_load_\w+ (v\[?\d+)", l)
                if not mm:
                    continue
                reg = mm.group(1)
                for k in range(i + 1, min(i + 6, len(ins))):
                    if ins[k].startswith("s_waitcnt vmcnt(0)") and k + 1 < len(ins) and ins[k + 1].startswith("ds_write") and reg in ins[k + 1]:
                        n += 1
                        break
                    if ins[k].startswith(("global_load", "buffer_load", "s_cbranch", "s_branch")):
                        break
            if n:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("uhdr::(anonymous namespace)::", "")
                print(f"{n:3d}  {os.path.basename(path):28s} {name[:120]}")
                worst = max(worst, n)
    return 1 if worst >= 7 else 0


if __name__ == "__main__":
    sys.exit(main())
