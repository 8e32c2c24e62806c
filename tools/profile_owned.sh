#!/bin/bash
# Kernel-trace stats of the sharded code path with the owner-keeps exchange in a world of one (no collectives issued),
# then the same bench unprofiled.  Outputs under gpurun_out/owned; the summary is copied into profiles/ by hand.
set -u
ROOT=/root/repo
OUT=$ROOT/gpurun_out/owned
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MP_BENCH_FORCE_SHARDED=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/trace.log 2>&1 || { echo "trace pass failed"; exit 1; }
echo "trace done"
cd $ROOT && timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_owned_world1.json 2> $OUT/bench_owned_world1.err || { echo "bench failed"; exit 1; }
echo "bench done"
find $OUT -name "*stats*.csv" | head
