#!/bin/bash
# PMC passes (one small counter group each) over the sharded path with the owner-keeps exchange in a world of one.
set -u
OUT=/root/repo/gpurun_out/pmc_owned
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MP_BENCH_FORCE_SHARDED=1
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o p$i -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
  echo "pass $i done"
done
find $OUT -name "*counter_collection.csv" | head
