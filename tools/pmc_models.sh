#!/bin/bash
# counters of k_propagate for the wide models (C3 bearings d = 4, C5 banded d = 16): tools/pmc_models.sh outdir   (GPU box, repo root)
OUT=$PWD/gpurun_out/${1:-pmc_models}; R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- python3 $R/tools/model_bench.py --steps 8 --which c3,c5 > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed ($grp)"; tail -n 3 $OUT/pmc$i.log; continue; }
done
cd $R && python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_propagate" in n or "k_draw_slots" in n:
            key = n.split("(")[0].replace("void ", "")[:48]
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    print("==", key)
    c = {k: sum(v[len(v)//2:]) / len(v[len(v)//2:]) for k, v in acc[key].items()}
    for k in sorted(c): print("   %-36s %16.1f" % (k, c[k]))
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        print("   -> VALU per wave %.0f, SALU per wave %.0f, LDS per wave %.0f" % (c.get("SQ_INSTS_VALU",0)/c["SQ_WAVES"], c.get("SQ_INSTS_SALU",0)/c["SQ_WAVES"], c.get("SQ_INSTS_LDS",0)/c["SQ_WAVES"]))
    if "TCC_REQ_sum" in c and c["TCC_REQ_sum"]:
        print("   -> L2 hit %.3f, EA reads %.0f" % (c["TCC_HIT_sum"]/c["TCC_REQ_sum"], c.get("TCC_EA0_RDREQ_sum",0)))
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        print("   -> wait/wave_cycles %.3f, active_valu/busy %.3f" % (c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_VALU"]/c["SQ_BUSY_CYCLES"]))
PY
