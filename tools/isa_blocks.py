#!/usr/bin/env python
"""Per-basic-block issue cost of one kernel (tools/isa_cost.py's model): python tools/isa_blocks.py <tu> <mangled-name-regex> [min_valu]"""
import collections, re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_cost import cost
tu, pat = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 10
text = open("/tmp/%s_isa.s" % tu).read()
m = re.search(r"^(_Z[^\n:]*%s[^\n:]*):\s*;[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel" % pat, text, re.S | re.M)
print(m.group(1))
blocks = []; cur = ("entry", "", [])
for l in m.group(2).split("\n"):
    mm = re.match(r"(\.LBB\d+_\d+):\s*;?(.*)", l)
    if mm: blocks.append(cur); cur = (mm.group(1), mm.group(2), [])
    else: cur[2].append(l)
blocks.append(cur)
tot = 0
for name, comment, ls in blocks:
    cnt = collections.Counter()
    for l in ls:
        mm = re.match(r"\s+([a-z][a-z_0-9]+)\s", l)
        if mm: cnt[mm.group(1)] += 1
    valu = sum(n for k, n in cnt.items() if k.startswith("v_")); units = sum(n * cost(k) for k, n in cnt.items())
    if "Loop" in comment: tot += units
    if valu >= minv:
        print("%-10s %-48s VALU %4d units %5.0f LDS %3d vmem %3d salu %3d wait %2d" % (name, comment[:48], valu, units, sum(n for k, n in cnt.items() if k.startswith("ds_")),
              sum(n for k, n in cnt.items() if k.startswith(("global_", "buffer_"))), sum(n for k, n in cnt.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop"))), cnt["s_waitcnt"]))
print("units in loop blocks: %.0f" % tot)
