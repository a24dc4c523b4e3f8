#!/usr/bin/env python
"""Kernels of libuhdr_hip.so that use scratch memory (.private_segment_fixed_size > 0 in the code objects' metadata), and the
register / LDS budget of every kernel.  The round-5 review counted sixteen spilling instantiations in the product binary; the rule
since round 6 is zero (tests/test_abi.py runs this).

    python tools/spill_check.py [--all]          --all: one line per kernel (vgprs, sgprs, LDS, scratch)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "libultrahdr_amd", "lib", "libuhdr_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib=LIB):
    """-> [{name, private_segment_fixed_size, vgpr_count, sgpr_count, group_segment_fixed_size}] over every gfx950 code object."""
    out = []
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "lib.so")
        shutil.copy(lib, so)
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+(?:- )?\.(name|private_segment_fixed_size|vgpr_count|sgpr_count|group_segment_fixed_size|agpr_count):\s+(\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k == "name" and not v.startswith("_Z") and not v.startswith("uhdr"):
                    continue  # an argument's .name
                if k in cur and k == "name":
                    pass
                cur[k] = v if k == "name" else int(v)
                if all(x in cur for x in ("name", "private_segment_fixed_size", "vgpr_count", "sgpr_count", "group_segment_fixed_size")):
                    out.append(cur)
                    cur = {}
    return out


if __name__ == "__main__":
    ks = kernels()
    spill = [k for k in ks if k["private_segment_fixed_size"] > 0]
    if "--all" in sys.argv:
        for k in sorted(ks, key=lambda k: k["name"]):
            name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
            print(f"{k['vgpr_count']:4d} vgpr {k['sgpr_count']:4d} sgpr {k['group_segment_fixed_size']:6d} B lds {k['private_segment_fixed_size']:4d} B scratch  {name[:150]}")
    print(f"{len(ks)} kernels, {len(spill)} with a private segment")
    for k in spill:
        print("  ", k["private_segment_fixed_size"], k["name"])
    sys.exit(1 if spill else 0)
