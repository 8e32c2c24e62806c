#!/usr/bin/env python3
"""Per-kernel summary of a tools/pmc_kernels.sh run: mean of every counter (second half of each kernel's launches = steady state),
mean duration from the --stats pass, and the ratios DESIGN.md quotes — VALU instructions per wave, issue utilisation, L2 hit
rate, and the MEASURED fp64 operation count: 64 lanes x (ADD + MUL + TRANS + 2 FMA) per wave-instruction (exec masks are not
seen by the SQ counters: an upper bound by the share of inactive lanes).

    python tools/pmc_kernels_summary.py gpurun_out/<dir> <kernel regex>"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("void ", "")
    depth, out = 0, []
    for ch in name:   # the kernel's name with its template arguments, without the parameter list
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out)


def main(root, regex):
    pat = re.compile(regex)
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    grid = {}
    for f in sorted(glob.glob(f"{root}/pmc_*/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if pat.search(k):
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                grid[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"]))
    dur = {}
    for f in sorted(glob.glob(f"{root}/trace/**/*_kernel_stats.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if pat.search(k):
                dur[k] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
    for k in sorted(set(vals) | set(dur)):
        print(f"== {k}")
        if k in grid:
            g = grid[k]
            print(f"   grid {g[0]} threads, workgroup {g[1]}, VGPRs {g[2]}, LDS {g[3]} B, scratch {g[4]} B")
        if k in dur:
            d = dur[k]
            print(f"   duration: mean {d[0]:.2f} us over {d[1]} launches (min {d[2]:.2f}, max {d[3]:.2f})")
        m = {}
        for c, v in vals.get(k, {}).items():
            v = v[len(v) // 2:]
            m[c] = sum(v) / len(v)
        for c in sorted(m):
            print(f"   {c:44s} {m[c]:18.1f}")
        w = m.get("SQ_WAVES", 0)
        if w:
            print(f"   -> per wave: VALU {m.get('SQ_INSTS_VALU', 0) / w:.1f}, SALU {m.get('SQ_INSTS_SALU', 0) / w:.1f}, LDS {m.get('SQ_INSTS_LDS', 0) / w:.1f}")
        if m.get("SQ_BUSY_CYCLES"):
            print(f"   -> SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES {m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_BUSY_CYCLES']:.3f}; "
                  f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES {m.get('SQ_WAIT_INST_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
        if "SQ_INSTS_VALU_FMA_F64" in m:
            ops = m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0) + 2 * m["SQ_INSTS_VALU_FMA_F64"]
            line = f"   -> fp64 operations per launch (64 lanes per wave-instruction, FMA = 2): {64 * ops:.4g}"
            if k in grid:
                line += f" = {64 * ops / grid[k][0]:.1f} per thread"
            if k in dur:
                line += f"; {64 * ops / (dur[k][0] * 1e-6) / 1e12:.2f} TFLOP/s at the mean duration"
            print(line)
        if m.get("TCC_REQ_sum"):
            print(f"   -> L2 hit rate {m.get('TCC_HIT_sum', 0) / m['TCC_REQ_sum']:.3f}; fabric reads (TCC_EA0_RDREQ) {m.get('TCC_EA0_RDREQ_sum', 0):.0f}")
        if m.get("TCP_TCC_READ_REQ_sum"):
            print(f"   -> mean L2 read latency seen by the TCP {m.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / m['TCP_TCC_READ_REQ_sum']:.1f} cycles")
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            print(f"   -> FETCH_SIZE {m.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} MB, WRITE_SIZE {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB per launch (KiB counters as reported)")
        print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ".")
