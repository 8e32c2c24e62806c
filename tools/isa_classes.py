#!/usr/bin/env python3
"""Static instruction mix of one kernel in a gfx950 .s file (hipcc -S --cuda-device-only), by issue class, priced with the
issue costs measured by tools/issue_cost.hip (profiles/r03/issue_costs.txt: cycles of SIMD time per wave-instruction at 8 waves
per SIMD).  Static counts: a loop body counts once (K1's loops: the polar retries, the forward row scan)."""
import collections
import re
import sys

COST = {"fp64 fma/mul/add": 2.9, "fp64 rcp/rsq/sqrt/div_*": 13.9, "v_mad_u64_u32": 3.5, "int VOP3 / 64-bit": 4.1, "int / move VOP1-2": 2.1,
        "cross-lane (dpp, permute, readlane)": 4.1, "LDS": 4.0, "global/flat memory": 4.0, "scalar ALU": 1.0, "scalar memory": 1.0, "branch / wait / misc": 1.0,
        "v_cmp / v_cndmask": 2.1}


def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_ldexp_f64", "v_fract", "v_rndne_f64", "v_floor_f64", "v_cvt_", "v_frexp", "v_trunc_f64")):
        return "fp64 fma/mul/add"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_")):
        return "fp64 rcp/rsq/sqrt/div_*"
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64"):
        return "v_mad_u64_u32"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "v_cmp / v_cndmask"
    if "dpp" in op or op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "ds_bpermute", "ds_permute", "ds_swizzle", "v_permlane", "v_mov_b32_dpp")):
        return "cross-lane (dpp, permute, readlane)"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "global/flat memory"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memrealtime", "s_memtime")):
        return "scalar memory"
    if op.startswith(("s_waitcnt", "s_branch", "s_cbranch", "s_barrier", "s_nop", "s_endpgm", "s_sleep", "s_setprio", "s_sethalt", "s_setreg", "s_getreg")):
        return "branch / wait / misc"
    if op.startswith("s_"):
        return "scalar ALU"
    if op.startswith(("v_lshl_add_u64", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_add3", "v_lshl_or", "v_and_or", "v_or3", "v_bfe", "v_bfi", "v_alignbit", "v_mul_lo",
                      "v_mul_hi", "v_mad_u32", "v_add_lshl", "v_lshl_add", "v_xad", "v_perm", "v_min3", "v_max3", "v_med3", "v_mbcnt")):
        return "int VOP3 / 64-bit"
    if op.startswith("v_"):
        return "int / move VOP1-2"
    return None


def main(path, needle):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and l.rstrip().endswith(":") is False and ":" in l)
    cnt = collections.Counter()
    ops = collections.Counter()
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end") or t.startswith("s_endpgm") and False:
            break
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r"^([a-z][a-z0-9_]+)\b", t)
        if not m or t.endswith(":"):
            continue
        c = classify(m.group(1))
        if c:
            cnt[c] += 1
            ops[m.group(1)] += 1
    tot = sum(cnt.values())
    print(f"kernel {needle}: {tot} static instructions")
    print(f"{'class':42s} {'count':>7s} {'share':>7s} {'issue cycles (static, per wave)':>32s}")
    cyc = 0.0
    for c, n in cnt.most_common():
        cy = n * COST[c]
        cyc += cy
        print(f"{c:42s} {n:7d} {100.0 * n / tot:6.1f}% {cy:32.0f}")
    print(f"{'total':42s} {tot:7d} {'':7s} {cyc:32.0f}")
    print("most frequent opcodes:", ", ".join(f"{o} {n}" for o, n in ops.most_common(14)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
