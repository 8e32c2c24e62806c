#!/bin/bash
# Round profile of the bench command (run on the GPU box from the repo root): kernel-trace stats, then FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md: one counter group per pass), SQ instruction counters, then
# the bench itself.  Outputs under gpurun_out/$1 (default r02_prof); the summaries are copied into profiles/rNN by
# tools/collect_profiles.sh on the build side.
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r03_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-sub-benches --no-systematic-leg"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 50 --warmup 10 > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -n 5 $OUT/trace.log; exit 1; }
echo "trace done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $B --steps 10 --warmup 3 > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed ($grp)"; tail -n 3 $OUT/pmc$i.log; continue; }
  echo "pmc pass $i done ($grp)"
done
# stamps of one launch in a busy queue (diagnostics build)
cd $R && timeout -k 10 200 python3 tools/stamp_probe.py --raw $OUT/stamps.npy > $OUT/stamps.json 2> $OUT/stamps.err && python3 tools/stamp_report.py $OUT/stamps.npy > $OUT/stamp_report.txt || echo "stamp probe failed"
cd /tmp
# the sharded code path in a world of one (owner-keeps exchange), kernel stats only
MP_BENCH_FORCE_SHARDED=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sharded -o bench -- $B --steps 50 --warmup 10 > $OUT/trace_sharded.log 2>&1 || echo "sharded trace pass failed"
cd $R && MP_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline > $OUT/bench_forced_sharded.json 2> $OUT/bench_forced_sharded.err || echo "forced-sharded bench failed"
cd $R && MP_BENCH_FORCE_SHARDED=1 MP_SHARD_ALWAYS_COLLECTIVE=1 timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline > $OUT/bench_forced_sharded_rccl.json 2> $OUT/bench_forced_sharded_rccl.err || echo "forced-sharded RCCL bench failed"
cd $R && timeout -k 10 300 python3 tools/route_scale.py > $OUT/route_scale.txt 2>&1 || echo "route_scale failed"
cd $R && timeout -k 10 200 python3 tools/mh_bench.py > $OUT/mh_functor_vs_handwritten.json 2> $OUT/mh_bench.err || echo "mh_bench failed"
cd $R && timeout -k 10 300 python3 tools/model_bench.py --which c3,c5,c4 > $OUT/model_bench.jsonl 2> $OUT/model_bench.err || echo "model_bench failed"
cd $R && timeout -k 10 600 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { echo "bench failed"; tail -n 5 $OUT/bench_n1.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_n1_k20.json 2>> $OUT/bench_n1.err || { echo "bench k20 failed"; exit 1; }
echo "bench done"
