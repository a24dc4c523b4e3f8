#!/usr/bin/env python
"""Instruction-class table of the hot loops of the issue-bound kernels (round-5 review: "no per-instruction-class table for any of
these kernels").  Compiles a translation unit to gfx950 ISA and, for every kernel whose mangled name matches a pattern, counts the
instructions of its LARGEST loop nest (the pixel / block loop) by class.

    python tools/isa_classes.py > profiles/r06_isa_classes.txt
    python tools/isa_classes.py tonemap 'tonemap_p010_kernelILi1ELb1E' 4        (TU, name regex, pixels per lane and iteration)
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "libultrahdr_amd", "csrc")

CLASSES = (
    ("convert  v_cvt_*", lambda o: o.startswith("v_cvt_")),
    ("select   v_cmp* / v_cndmask", lambda o: o.startswith(("v_cmp", "v_cndmask"))),
    ("clamp    v_med3 / v_min / v_max", lambda o: o.startswith(("v_med3", "v_min", "v_max"))),
    ("f64      v_*_f64", lambda o: o.endswith("_f64") and o.startswith("v_")),
    ("f32 mul  v_mul_f32 / v_fma / v_mac / v_pk_mul", lambda o: o.startswith(("v_mul_f32", "v_fma", "v_mac", "v_pk_mul", "v_pk_fma", "v_mul_legacy", "v_fmac"))),
    ("f32 add  v_add_f32 / v_sub_f32 / v_pk_add", lambda o: o.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_add_f32"))),
    ("f32 other (rcp, rndne, floor, frexp, ldexp, ...)", lambda o: o.startswith("v_") and o.endswith(("_f32", "_f16")) ),
    ("int mul / mad", lambda o: o.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i", "v_mul_u32", "v_mul_i32", "v_mad_u64"))),
    ("int shift / bit field / perm", lambda o: o.startswith(("v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_bfi", "v_perm", "v_alignb", "v_and", "v_or", "v_xor", "v_not", "v_lshl_or", "v_and_or", "v_or3", "v_bfm", "v_ffb", "v_bcnt", "v_sad"))),
    ("int add / sub", lambda o: o.startswith(("v_add_u", "v_add_co", "v_addc", "v_sub_u", "v_sub_co", "v_subb", "v_subrev_u", "v_subrev_co", "v_add3", "v_add_i", "v_sub_i", "v_add_nc", "v_sub_nc", "v_lshl_add", "v_add_lshl"))),
    ("move / lane  v_mov / v_readlane / v_writelane / dpp / swap", lambda o: o.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap", "v_accvgpr", "v_permlane"))),
    ("LDS read   ds_read* / ds_bpermute", lambda o: o.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle"))),
    ("LDS write  ds_write*", lambda o: o.startswith(("ds_write", "ds_add", "ds_or", "ds_max", "ds_min"))),
    ("global load", lambda o: o.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))),
    ("global store / atomic", lambda o: o.startswith(("global_store", "buffer_store", "flat_store", "global_atomic", "scratch_store", "buffer_atomic"))),
    ("scalar ALU / branch  s_*", lambda o: o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer_load"))),
    ("scalar load  s_load*", lambda o: o.startswith(("s_load", "s_buffer_load"))),
    ("wait / nop  s_waitcnt / s_nop", lambda o: o.startswith(("s_waitcnt", "s_nop"))),
)


def isa_of(tu):
    out = f"/tmp/{tu}_isa.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fwrapv", "-fvisibility=hidden",
                           "-I" + SRC, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(SRC, tu + ".hip"), "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(text, pat):
    for m in re.finditer(r"^(_Z[^\n:]*(?:%s)[^\n:]*):\s*;[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel" % pat, text, re.S | re.M):
        yield m.group(1), m.group(2)


def largest_loop(body):
    lines = body.split("\n")
    spans = []
    for i, l in enumerate(lines):
        mm = re.match(r"\.LBB(\d+_\d+):.*Loop Header: Depth=1", l)
        if mm:
            tag = "Header=BB" + mm.group(1) + " "
            inside = [j for j, x in enumerate(lines) if tag in x]
            end = max(inside) if inside else i
            # nested loops name their own header: extend to the last block that is "in Loop: Header=BB<this>" at any depth
            while end + 1 < len(lines) and not re.match(r"\.LBB\d+_\d+:", lines[end + 1]):
                end += 1
            spans.append((end - i, i, end))
    if not spans:
        return lines
    _, a, b = max(spans)
    return lines[a:b + 1]


def table(name, lines, px):
    cnt = collections.Counter()
    total = 0
    for l in lines:
        mm = re.match(r"\s+([a-z][a-z_0-9]+)(\s|$)", l)
        if not mm:
            continue
        op = mm.group(1)
        for cname, test in CLASSES:
            if test(op):
                cnt[cname] += 1
                break
        else:
            cnt["other  " + op] += 1
        total += 1
    valu = sum(n for c, n in cnt.items() if not c.startswith(("LDS", "global", "scalar", "wait", "other")))
    print(f"== {name}")
    print(f"   largest loop nest: {total} instructions, {valu} VALU" + (f" = {valu / px:.1f} VALU per pixel ({px} pixels per lane and iteration, static count: every branch of the loop body counted once)" if px else ""))
    for c, n in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print(f"   {n:5d}  {100.0 * n / total:5.1f} %  {c}")
    print()


DEFAULT = (
    # TU, mangled-name regex, pixels per lane per iteration (None: not a per-pixel loop), label
    ("tonemap", "tonemap_p010_kernelILi1ELb1E", 4, "tonemap_p010_kernel<GAMUT=1, LUT=true>  (4K P010 HLG -> YCbCr 4:2:0, the bench's tonemap_4k_p010)"),
    ("encode_fused", "encode_api0_fused4_kernelILb0ELi2ELi1ELi1E", 4, "encode_api0_fused4_kernel<one pass, HDR-side gamut, 3 ch, tone-map gamut>  (config 3: 8K RGBA1010102 PQ)"),
    ("encode_api1_fused", "base_blocks_kernel", None, "base_blocks_kernel  (convertYuv + 3 x FDCT + quantize of the base image; a wave = 64 blocks' rows)"),
    ("apply_gainmap", "apply_quad_kernel_s96ILi1ELi0ELi1ELi0ELi0E", 8, "apply_quad_kernel_s96<HLG out, Y400 map, scale 4>  (config 5)"),
    ("apply_gainmap", "apply_quad_kernelILi0ELi2ELi0ELi0ELi0E", 8, "apply_quad_kernel<F16 out, RGBA8888 map, scale 1>  (north star, for comparison)"),
)

if __name__ == "__main__":
    jobs = DEFAULT if len(sys.argv) < 3 else ((sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "0" else None, sys.argv[2]),)
    cache = {}
    for tu, pat, px, label in jobs:
        if tu not in cache:
            cache[tu] = isa_of(tu)
        found = False
        for name, body in kernels(cache[tu], pat):
            table(label + "\n   " + name, largest_loop(body), px)
            found = True
            break
        if not found:
            print("== " + label + ": kernel not found (" + pat + ")\n")
