#!/bin/bash
# SQ counters for the bench command: one small counter group per pass (rocprofv3 --pmc), each pass bounded.
set -u
OUT=/root/repo/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o p$i -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
  echo "pass $i done"
done
ls $OUT
