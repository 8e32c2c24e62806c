#!/bin/bash
# Counters + durations of the kernels one command launches, kernel by kernel (GPU box, repo root):
#   tools/pmc_kernels.sh OUTDIR "GROUPS" KERNEL_REGEX python3-script [args...]
# GROUPS: a subset of "inst busy f64 tcc fetch write tcp" (one rocprofv3 --pmc pass each: MI355X_MICROARCH.md, one counter group per
# pass), plus one --kernel-trace --stats pass for the durations.  The script is run as `python3 <script> <args>` directly behind
# `--` (no env / bash hop: the profiler's library has initialised the GPU by then).  Summary: OUTDIR/summary.txt.
OUT=$PWD/gpurun_out/$1; GROUPS_="$2"; REGEX="$3"; shift 3
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SCRIPT=$R/$1; shift
declare -A G
G[inst]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS"
G[busy]="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
G[f64]="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
G[tcc]="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
G[tcp]="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $SCRIPT "$@" > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -n 5 $OUT/trace.log; }
echo "trace pass done"
for g in $GROUPS_; do
  timeout -k 10 300 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $OUT/pmc_$g -o p -- python3 $SCRIPT "$@" > $OUT/pmc_$g.log 2>&1 || { echo "pmc pass $g failed"; tail -n 3 $OUT/pmc_$g.log; continue; }
  echo "pmc pass $g done"
done
cd $R && python3 tools/pmc_kernels_summary.py $OUT "$REGEX" > $OUT/summary.txt 2>&1
tail -n 40 $OUT/summary.txt
