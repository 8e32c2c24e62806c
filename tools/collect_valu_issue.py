#!/usr/bin/env python3
"""VALU-issue fraction of the wide models' propagate kernels (C3 bearings d = 4, C5 banded / dense d = 16) from a tools/pmc_kernels.sh
run of tools/model_bench.py / tools/dense_bench.py, stamped with the content hash of the sources the library was built from:
bench.py's c3 / c5_shard / c5_dense_shard objects report it next to their bytes-based fraction — these kernels are bound by
instruction issue (16 polar normals per particle at d = 16), and a bytes / s figure alone measures the wrong resource.

    valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x mean launch duration x 2.4 GHz)

(a wave64 instruction occupies a 16-lane SIMD for 4 cycles; 2.4 GHz is the card's peak engine clock: the clock it actually holds under
these kernels is lower — 2.05 - 2.2 GHz by the stamps —, so the figure is a LOWER bound of the share of issue slots in use).

    python tools/collect_valu_issue.py gpurun_out/r05_prof/pmc_models gpurun_out/r05_prof/pmc_dense > profiles/r05/valu_issue.json"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from modppl_amd import build as B  # noqa: E402
from tools.pmc_kernels_summary import short  # noqa: E402

SIMDS, CYCLES_PER_INST, CLOCK_HZ = 1024, 4.0, 2.4e9
out = {"_measured": {"source_hash": B.source_hash(), "formula": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x mean launch duration x 2.4 GHz): a lower bound (the held clock is lower)",
                     "from": "tools/pmc_kernels.sh <dir> 'inst busy ...' 'k_propagate|k_draw_slots' tools/model_bench.py --steps 8 --which c3,mid,c5 (and tools/dense_bench.py)"}}
for root in sys.argv[1:]:
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{root}/pmc_*/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for f in sorted(glob.glob(f"{root}/trace/**/*_kernel_stats.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            dur[short(r["Name"])] = float(r["AverageNs"]) * 1e-9
    for k, v in vals.items():
        if "k_propagate" not in k or "SQ_INSTS_VALU" not in v or k not in dur:
            continue
        x = v["SQ_INSTS_VALU"]
        insts = sum(x[len(x) // 2:]) / len(x[len(x) // 2:])
        rec = {"valu_wave_instructions_per_launch": insts, "mean_launch_us": dur[k] * 1e6,
               "valu_issue_frac": insts * CYCLES_PER_INST / (SIMDS * dur[k] * CLOCK_HZ)}
        if "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v:
            a, b = v["SQ_ACTIVE_INST_VALU"], v["SQ_BUSY_CYCLES"]
            rec["sq_active_inst_valu_over_busy_cycles"] = (sum(a[len(a) // 2:]) / len(a[len(a) // 2:])) / (sum(b[len(b) // 2:]) / len(b[len(b) // 2:]))
        out[k] = rec
print(json.dumps(out, indent=1))
