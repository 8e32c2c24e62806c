#!/usr/bin/env python3
"""Static instruction mix of k_propagate<mp_lgssm1, 1024> per PHASE: the -DMP_STAMPS build of the kernel carries an s_memtime at
every phase boundary (MP_STAMP(0, slot, 0) in mp_pf_kernels.h; the stamp's store has offset slot * 8), so the .s splits there.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DMP_STAMPS -S --cuda-device-only modppl_amd/csrc/mp_pf.hip -o /tmp/s.s
    python tools/k1_phases.py /tmp/s.s
Code order is layout order, not execution order: a phase's cold branches (the polar retry loop, the non-fused path) sit in the
segment where the compiler laid them out.  Loop bodies count once."""
import collections
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_classes import COST, classify  # noqa: E402

NAMES = {0: "start", 16: "table: loads out", 17: "table: level 1 done", 18: "table: scan done", 19: "table in LDS", 25: "draws: Philox + target", 26: "draws: tile located",
         27: "draws: guide cell read", 28: "draws: row pair read", 29: "draws: forward scan done", 20: "lookups done, parents out", 21: "deviates done",
         22: "model + weights done", 2: "particles done", 8: "normalise: tile max", 9: "normalise: exp + quantise", 10: "normalise: scan", 11: "rows stored",
         12: "guide zeroed", 7: "tile scalars out", 13: "guide stored", 14: "ticket", 15: "tail", 24: "x_out", 3: "normalise done", 4: "end"}


def main(path):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z11k_propagateI9mp_lgssm1Li1024E") and ":" in l)
    seg = collections.OrderedDict()
    cur = "entry"
    seg[cur] = collections.Counter()
    pending = False
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r"^([a-z][a-z0-9_]+)\b", t)
        if not m or t.endswith(":"):
            continue
        op = m.group(1)
        if op == "s_memtime":
            pending = True
            continue
        if pending and op.startswith(("global_store_dwordx2", "flat_store_dwordx2")):
            mo = re.search(r"offset:(\d+)", t)
            slot = int(mo.group(1)) // 8 if mo else 0
            cur = f"after stamp {slot:2d} ({NAMES.get(slot, '?')})"
            seg.setdefault(cur, collections.Counter())
            pending = False
            continue
        c = classify(op)
        if c:
            seg[cur][c] += 1
    classes = ["fp64 fma/mul/add", "fp64 rcp/rsq/sqrt/div_*", "v_mad_u64_u32", "int VOP3 / 64-bit", "int / move VOP1-2", "v_cmp / v_cndmask",
               "cross-lane (dpp, permute, readlane)", "LDS", "global/flat memory", "scalar ALU"]
    short = ["f64", "f64 slow", "mad64", "int3", "int", "cmp/sel", "xlane", "LDS", "vmem", "salu"]
    print(f"{'segment (layout order)':52s}" + "".join(f"{s:>9s}" for s in short) + f"{'VALU cyc':>10s}")
    for name, cnt in seg.items():
        if sum(cnt.values()) < 8:
            continue
        cyc = sum(cnt[c] * COST[c] for c in classes[:7])
        print(f"{name:52s}" + "".join(f"{cnt[c]:9d}" for c in classes) + f"{cyc:10.0f}")


if __name__ == "__main__":
    main(sys.argv[1])
