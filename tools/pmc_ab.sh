#!/bin/bash
# counters of the headline step for one build of the library: tools/pmc_ab.sh lib.so outdir   (run on the GPU box from the repo root)
L=$PWD/$1; OUT=$PWD/gpurun_out/$2; R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MODPPL_HIP_LIB=$L
B="python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-sub-benches --no-systematic-leg --steps 10 --warmup 3"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $B > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed ($grp)"; tail -n 3 $OUT/pmc$i.log; continue; }
done
cd $R && python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc*/**/p_counter_collection.csv", recursive=True) + glob.glob("$OUT/pmc*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_propagate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k][len(acc[k]) // 2:]   # later launches: steady state
    print("%-36s %14.1f" % (k, sum(v) / len(v)))
PY
