#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --kernel-trace run of tools/route_scale.py, split by world (the table kernel's grid tells)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    if "shard" not in k:
        continue
    g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r["Workgroup_Size"])
    agg.setdefault((k, g // wg), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in agg.items():
    print(f"{k:46s} workgroups {g:>6d}  launches {len(v):4d}  avg {sum(v) / len(v):7.2f} us")
