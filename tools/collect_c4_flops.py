#!/usr/bin/env python3
"""MEASURED fp64 operation count per chain-iteration of the regen-MH kernels (BASELINE configs[3]) from a tools/pmc_kernels.sh
run of tools/mh_bench.py: 64 lanes x (ADD + MUL + TRANS + 2 FMA) wave-instructions per launch / (chains x iterations per launch),
stamped with the content hash of the sources the library was built from — bench.py's c4.roofline takes its flop count from here.

    python tools/collect_c4_flops.py gpurun_out/r04_prof/pmc_mh 30 > profiles/r04/c4_flops.json"""
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

root, iters = sys.argv[1], int(sys.argv[2])
vals = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for f in sorted(glob.glob(f"{root}/pmc_*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        grid[k] = int(r["Grid_Size"])
out = {"_measured": {"source_hash": B.source_hash(), "iterations_per_launch": iters, "from": "tools/pmc_kernels.sh <dir> 'inst busy f64' 'k_mh|k_fn' tools/mh_bench.py 1048576 %d" % iters,
                     "formula": "64 x (SQ_INSTS_VALU_ADD_F64 + _MUL_F64 + _TRANS_F64 + 2 x _FMA_F64) / chains / iterations (wave-instruction counts: inactive lanes included)"}}
for k in ("k_mh_iterate<0>", "k_fn_regen<mp_hier_fn>", "k_mh_iterate<1>", "k_fn_mh<mp_hier_fn, mp_hier_drift_fn>"):
    if k not in vals or "SQ_INSTS_VALU_FMA_F64" not in vals[k]:
        continue
    m = {c: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for c, v in vals[k].items()}
    ops = 64.0 * (m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0) + 2 * m["SQ_INSTS_VALU_FMA_F64"])
    out[k] = {"flop_per_chain_iteration": ops / grid[k] / iters, "valu_instructions_per_chain_iteration_per_wave": m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_WAVES", 1), 1) / iters,
              "fp64_arithmetic_share_of_valu": (m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0) + m["SQ_INSTS_VALU_FMA_F64"]) / max(m.get("SQ_INSTS_VALU", 1), 1)}
print(json.dumps(out, indent=1))
