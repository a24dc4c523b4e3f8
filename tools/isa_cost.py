#!/usr/bin/env python
"""Static issue-cost estimate of a kernel TU (round 4): compiles csrc/<tu>.hip to gfx950 ISA and prints, per kernel, registers /
LDS / occupancy and the instruction mix weighted with the issue costs tools/ubench9 measured (v_fma_f32 = 1).
    python tools/isa_cost.py tonemap [kernel-name-substring]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "libultrahdr_amd", "csrc")
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_mov_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_mac_f32", "v_not_b32", "v_ashrrev_i32"}
TRANS = {"v_rcp_f32", "v_log_f32", "v_exp_f32", "v_sqrt_f32", "v_rsq_f32", "v_rcp_iflag_f32"}
def cost(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if not base.startswith("v_"): return 0.0
    if base in TRANS: return 3.2
    if base in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"): return 6.4
    if base.endswith("_f64") or base.startswith("v_pk_") or "f64" in base: return 1.85
    if base in FAST and not op.endswith("_sdwa"): return 1.0
    return 1.7
def main():
    tu = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
    out = "/tmp/%s_isa.s" % tu
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fwrapv", "-fvisibility=hidden",
                           "-I" + SRC, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(SRC, tu + ".hip"), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    for m in re.finditer(r"^(_Z[^\n:]*):\s*;[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pat and pat not in dem: continue
        cnt = collections.Counter()
        for l in body.split("\n"):
            mm = re.match(r"\s+([a-z][a-z_0-9]+)\s", l)
            if mm: cnt[mm.group(1)] += 1
        valu = sum(n for k, n in cnt.items() if k.startswith("v_"))
        units = sum(n * cost(k) for k, n in cnt.items())
        info = re.findall(r";\s*(NumVgprs|NumAgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize):\s*(\d+)", text[m.end():m.end() + 6000])
        print("%s\n   %s" % (dem[:150], " ".join("%s=%s" % kv for kv in info[:6])))
        print("   static: %d VALU instr, %.0f issue units; LDS %d, VMEM %d, SALU %d, branches %d" % (
            valu, units, sum(n for k, n in cnt.items() if k.startswith("ds_")), sum(n for k, n in cnt.items() if k.startswith(("global_", "buffer_", "flat_"))),
            sum(n for k, n in cnt.items() if k.startswith("s_") and not k.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop"))),
            sum(n for k, n in cnt.items() if k.startswith(("s_cbranch", "s_branch")))))
        top = sorted(cnt.items(), key=lambda kv: -kv[1] * max(cost(kv[0]), 0.01))[:14]
        print("   " + ", ".join("%s x%d" % kv for kv in top))
if __name__ == "__main__":
    main()
