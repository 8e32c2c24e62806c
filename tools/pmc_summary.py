#!/usr/bin/env python3
"""Per-kernel means of every counter collected by tools/profile_round.sh (one rocprofv3 --pmc group per pass), plus the ratios
DESIGN.md quotes: VALU wave-instructions per wave and per unit of work, issue utilisation, L2 hit rate, mean L2 read latency.

    python tools/pmc_summary.py gpurun_out/r02_prof > profiles/r02/counters_summary.txt"""
import collections
import csv
import glob
import sys

KERNELS = ["k_propagate_mt<mp_lgssm1", "k_propagate<mp_lgssm1", "k_draw_slots", "k_resolve_slots", "k_shard_own_draw", "k_shard_own_place", "k_shard_own_plan", "k_shard_table", "k_build_table"]
UNITS = 1 << 20   # particles (draws) per launch of the bench workload


def main(root):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{root}/pmc*/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            for k in KERNELS:
                if k in name and int(r.get("Grid_Size", "0") or 0) >= (1 << 18 if "plan" not in k and "table" not in k else 0):
                    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in KERNELS:
        if k not in vals:
            continue
        m = {c: sum(v) / len(v) for c, v in vals[k].items()}
        print(f"== {k}   (launches sampled: {max(len(v) for v in vals[k].values())})")
        for c in sorted(m):
            print(f"   {c:44s} {m[c]:16.1f}")
        if "SQ_WAVES" in m and m["SQ_WAVES"]:
            w = m["SQ_WAVES"]
            print(f"   -> VALU wave-instructions per wave {m.get('SQ_INSTS_VALU', 0) / w:8.1f} = {m.get('SQ_INSTS_VALU', 0) * 64 / UNITS:7.1f} lane-operations per particle/draw;"
                  f" SALU per wave {m.get('SQ_INSTS_SALU', 0) / w:7.1f}; LDS per wave {m.get('SQ_INSTS_LDS', 0) / w:6.1f}")
        if "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
            print(f"   -> SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES {m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_BUSY_CYCLES']:6.3f}; "
                  f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES {m.get('SQ_WAIT_INST_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):6.3f}")
        if "TCC_REQ_sum" in m and m["TCC_REQ_sum"]:
            print(f"   -> L2 hit rate {m.get('TCC_HIT_sum', 0) / m['TCC_REQ_sum']:6.3f} ({m['TCC_REQ_sum'] / UNITS:5.2f} L2 requests per particle/draw, {m.get('TCC_MISS_sum', 0) / UNITS:5.2f} misses)")
        if "TCP_TCC_READ_REQ_sum" in m and m["TCP_TCC_READ_REQ_sum"]:
            print(f"   -> mean L2 read latency seen by the TCP {m.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / m['TCP_TCC_READ_REQ_sum']:7.1f} cycles over "
                  f"{m['TCP_TCC_READ_REQ_sum'] / UNITS:5.2f} read requests per particle/draw")
        print()


if __name__ == "__main__":
    main(sys.argv[1])
