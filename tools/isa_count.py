#!/usr/bin/env python
"""Static instruction mix of the apply_quad_kernel main loops (round 3: the kernel is VALU-issue bound, so the VALU count
per pixel is the number to drive down).  Compiles apply_gainmap.hip to ISA and counts, per variant, the instructions of
the innermost loop (8 pixels per lane per iteration: two quads), leaving out the sub-normal half slow paths (blocks entered
through `s_cbranch_vccz` after the v_cmp of the fast-path test) -- both gamut branches are counted (one executes).

    python tools/isa_count.py [variant ...]      variant = template arguments, e.g. 0,0,1,0,0 (map A) 0,2,0,0,0 (map C)
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "libultrahdr_amd", "csrc")


def main():
    variants = sys.argv[1:] or ["0,0,1,0,0", "0,2,0,0,0", "1,0,1,0,0"]
    out = "/tmp/apply_isa.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fwrapv",
                           "-fvisibility=hidden", "-I" + SRC, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                           os.path.join(SRC, "apply_gainmap.hip"), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    for v in variants:
        mang = "apply_quad_kernel(?:_s96)?I" + "".join("Li%sE" % a for a in v.split(",")) + "EEv"
        m = re.search(r"^(_ZN4uhdr[^\n]*%s[^\n]*):\s*;[^\n]*\n(.*?)\n\s*\.amdhsa_kernel" % mang, text, re.S | re.M)
        if not m:
            print(v, "not found")
            continue
        lines = m.group(2).split("\n")
        # the main loop = the loop header whose "in Loop: Header=..." blocks span the most lines
        spans = []
        for i, l in enumerate(lines):
            mm = re.match(r"\.LBB(\d+_\d+):.*Loop Header", l)
            if mm:
                tag = "Header=BB" + mm.group(1) + " "
                inside = [j for j, x in enumerate(lines) if tag in x]
                if inside:
                    end = max(inside)
                    while end + 1 < len(lines) and not re.match(r"\.LBB\d+_\d+:", lines[end + 1]):
                        end += 1
                    spans.append((end - min(i, min(inside)), min(i, min(inside)), end))
        if not spans:
            print(v, "no loop")
            continue
        _, a, b = max(spans)
        # basic blocks of the loop; the sub-normal half slow path (recognised by its 2^25 multiplications) is left out
        blocks, cur = [], []
        for l in lines[a:b + 1]:
            if re.match(r"\.LBB\d+_\d+:", l.strip()) or re.match(r"; %bb\.\d+:", l.strip()):
                blocks.append(cur)
                cur = []
            cur.append(l)
        blocks.append(cur)
        body = [l for blk in blocks if not any("0x4c000000" in x for x in blk) for l in blk]
        cnt = collections.Counter()
        for l in body:
            mm = re.match(r"\s+([a-z_0-9]+)", l)
            if mm and not l.strip().startswith(";"):
                cnt[mm.group(1)] += 1
        grp = lambda p: sum(n for k, n in cnt.items() if k.startswith(p))
        vg = re.search(r"\.set %s[^\n]*\.num_vgpr, (\d+)" % re.escape(m.group(1)), text)
        sg = re.search(r"\.set %s[^\n]*\.numbered_sgpr, (\d+)" % re.escape(m.group(1)), text)
        print(f"<{v}>: VALU {grp('v_')}  SALU {grp('s_')}  DS {grp('ds_')}  VMEM {grp('global_') + grp('buffer_')}   (per 8 pixels; vgpr {vg.group(1) if vg else '?'} sgpr {sg.group(1) if sg else '?'})")
        print("   " + "  ".join(f"{k} {n}" for k, n in cnt.most_common(28) if k.startswith(("v_", "ds_", "global_", "buffer_"))))


if __name__ == "__main__":
    main()
