#!/bin/bash
# Round profile of the bench command: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md: one counter group per pass), each bounded.  Outputs under gpurun_out/r1; summaries are copied
# into profiles/ by hand (tools/collect_traffic.py for the PMC passes).
set -u
ROOT=/root/repo
OUT=$ROOT/gpurun_out/r1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/trace.log 2>&1 || { echo "trace pass failed"; exit 1; }
echo "trace done"
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
echo "fetch done"
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/write.log 2>&1 || { echo "write pass failed"; exit 1; }
echo "write done"
cd $ROOT && timeout -k 10 300 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { echo "bench failed"; exit 1; }
echo "bench done"
find $OUT -name "*.csv" | head -20
