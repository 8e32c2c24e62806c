#!/bin/bash
# Round profile of the bench command (run on the GPU box from the repo root): kernel-trace stats, then FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md: one counter group per pass), SQ instruction counters, then
# the bench itself.  Outputs under gpurun_out/$1 (default r02_prof); the summaries are copied into profiles/rNN by
# tools/collect_profiles.sh on the build side.
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r05_prof}
DIAG=$R/modppl_amd/csrc/libmodppl_hip_diag.so   # the A/B switches below exist in the diagnostics build only (csrc/mp_diag.h)
mkdir -p $OUT
# the content hash of the sources the library on THIS box was built from: tools/collect_profiles.sh refuses a run of another tree
(cd $R && python3 -c "from modppl_amd import build as B; print(B.source_hash())") > $OUT/source_hash.txt
PART=${2:-all}   # core | sharded | pmc | all: one gpurun call allows 20 minutes, the whole round takes about fifty
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-sub-benches --no-systematic-leg"
if [ "$PART" = core ] || [ "$PART" = all ]; then
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 50 --warmup 10 > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -n 5 $OUT/trace.log; exit 1; }
echo "trace done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $B --steps 10 --warmup 3 > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed ($grp)"; tail -n 3 $OUT/pmc$i.log; continue; }
  echo "pmc pass $i done ($grp)"
done
# stamps of one launch in a busy queue (diagnostics build): the two-tiles-per-workgroup kernel (the default at 2^20 particles) and,
# with MP_K1_MT=0, the one-workgroup-per-tile kernel
cd $R && timeout -k 10 200 python3 tools/stamp_probe.py --raw $OUT/stamps.npy > $OUT/stamps.json 2> $OUT/stamps.err && python3 tools/stamp_report_mt.py $OUT/stamps.npy > $OUT/stamp_report.txt || echo "stamp probe failed"
cd $R && MP_K1_MT=0 timeout -k 10 200 python3 tools/stamp_probe.py --raw $OUT/stamps_tile.npy > $OUT/stamps_tile.json 2> $OUT/stamps_tile.err && python3 tools/stamp_report.py $OUT/stamps_tile.npy > $OUT/stamp_report_tile_kernel.txt || echo "stamp probe (tile kernel) failed"
# one tile pass against a tile's share of a CU, both forms of K1
cd $R && { echo "# k_propagate_mt from two tiles per CU (default)"; timeout -k 10 300 python3 tools/k1_scaling.py; echo "# MP_K1_MT=0: k_propagate at every size"; MODPPL_HIP_LIB=$DIAG MP_K1_MT=0 timeout -k 10 300 python3 tools/k1_scaling.py; } > $OUT/k1_scaling.txt 2>/dev/null || echo "k1_scaling failed"
# the same K = 200 bench with either form of K1, alternating (same box)
cd $R && for i in 1 2 3; do for v in 1 0; do MODPPL_HIP_LIB=$DIAG MP_K1_MT=$v timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline --no-systematic-leg --repeats 3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('MP_K1_MT=$v', round(d['ms_per_step'] * 1e3, 2), 'us per step;', d['roofline']['kernel'], round(d['roofline']['kernel_us'], 2), 'us')"; done; done > $OUT/k1_forms_ab.txt || echo "k1 forms A/B failed"
# the reference-shaped (synchronous) loop with and without the host-mapped mirror (MP_HOST_MIRROR=0: k_draw_slots + k_resolve_slots + a copy of the scalars per step, as in round 3)
cd $R && for v in 1 0; do MODPPL_HIP_LIB=$DIAG MP_HOST_MIRROR=$v timeout -k 10 300 python3 -c "
import json, bench, modppl_amd
ys = bench.lgssm_observations(80)
r = bench.reference_shaped_loop(modppl_amd.lgssm_model(*bench.LGSSM_PARAMS), 1 << 20, ys, 50, 5)
r['MP_HOST_MIRROR'] = $v
print(json.dumps(r))"; done > $OUT/reference_shaped_loop.jsonl 2>/dev/null || echo "reference-shaped loop failed"
# ... and the same loop from compiled host code (tools/sync_loop.cpp over the C++ wrapper): without the interpreter's share
cd $R && g++ -std=c++17 -O2 tools/sync_loop.cpp -o $OUT/sync_loop -Lmodppl_amd/csrc -lmodppl_hip -Wl,-rpath,$R/modppl_amd/csrc 2> $OUT/sync_loop.err && { timeout -k 10 120 $OUT/sync_loop; } > $OUT/reference_shaped_loop_cpp.txt 2>&1 || echo "sync_loop (C++) failed"
rm -f $OUT/sync_loop
cd /tmp
# the reference-shaped loop again, three times over (tools/sync_probe.py): the spread from run to run
cd $R && timeout -k 10 200 python3 tools/sync_probe.py > $OUT/sync_probe.jsonl 2>/dev/null || echo "sync_probe failed"
cd $R && timeout -k 10 600 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { echo "bench failed"; tail -n 5 $OUT/bench_n1.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_n1_k20.json 2>> $OUT/bench_n1.err || { echo "bench k20 failed"; exit 1; }
fi
if [ "$PART" = sharded ] || [ "$PART" = all ]; then
cd /tmp
# the sharded code path in a world of one (owner-keeps exchange), kernel stats only
MP_BENCH_FORCE_SHARDED=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sharded -o bench -- $B --steps 50 --warmup 10 > $OUT/trace_sharded.log 2>&1 || echo "sharded trace pass failed"
MP_BENCH_FORCE_SHARDED=1 MP_SHARD_EXCHANGE=split timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sharded_split -o bench -- $B --steps 50 --warmup 10 > $OUT/trace_sharded_split.log 2>&1 || echo "sharded split trace pass failed"
cd $R && MP_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline > $OUT/bench_forced_sharded.json 2> $OUT/bench_forced_sharded.err || echo "forced-sharded bench failed"
cd $R && MP_BENCH_FORCE_SHARDED=1 MP_SHARD_ALWAYS_COLLECTIVE=1 timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline > $OUT/bench_forced_sharded_rccl.json 2> $OUT/bench_forced_sharded_rccl.err || echo "forced-sharded RCCL bench failed"
cd $R && MP_BENCH_FORCE_SHARDED=1 MP_SHARD_EXCHANGE=split timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline --no-systematic-leg > $OUT/bench_forced_sharded_split.json 2> $OUT/bench_forced_sharded_split.err || echo "forced-sharded split bench failed"
cd $R && timeout -k 10 300 python3 tools/route_scale.py > $OUT/route_scale.txt 2>&1 || echo "route_scale failed"
# the timeline of one sharded resample + step on one clock (-DMP_STAMPS build): the table launch's phases, the host's gap, the propagate kernel
cd $R && timeout -k 10 200 python3 tools/table_stamps.py > $OUT/table_stamps.txt 2>&1 || echo "table_stamps failed"
cd $R && timeout -k 10 200 python3 tools/table_stamps.py --two-calls --schemes 3 > $OUT/table_stamps_two_calls.txt 2>&1 || echo "table_stamps (two calls) failed"
cd $R && timeout -k 10 200 python3 tools/mh_bench.py > $OUT/mh_functor_vs_handwritten.json 2> $OUT/mh_bench.err || echo "mh_bench failed"
cd $R && timeout -k 10 300 python3 tools/model_bench.py --which c3,mid,c5,c4 > $OUT/model_bench.jsonl 2> $OUT/model_bench.err || echo "model_bench failed"
fi
if [ "$PART" = pmc ] || [ "$PART" = all ]; then
# counters, fp64 operation counts and durations of the other configurations' kernels (C3 / C5 propagate kernels, the MH kernels)
cd $R && bash tools/pmc_kernels.sh $(basename $OUT)/pmc_models "inst busy f64 tcc fetch write tcp" "k_propagate|k_draw_slots" tools/model_bench.py --steps 8 --which c3,mid,c5 > $OUT/pmc_models.log 2>&1 || echo "pmc_models failed"
cd $R && bash tools/pmc_kernels.sh $(basename $OUT)/pmc_mh "inst busy f64" "k_mh|k_fn" tools/mh_bench.py 1048576 30 > $OUT/pmc_mh.log 2>&1 || echo "pmc_mh failed"
# the sharded resample's kernels (one rank of emulated worlds of 1 .. 8): table + counts + plan, placement, the propagate kernel's SHD form
cd $R && bash tools/pmc_kernels.sh $(basename $OUT)/pmc_sharded "inst busy tcc" "k_shard_table|k_shard_self|k_shard_own|k_propagate" tools/route_scale.py > $OUT/pmc_sharded.log 2>&1 || echo "pmc_sharded failed"
cd $R && bash tools/pmc_kernels.sh $(basename $OUT)/pmc_dense "inst busy f64 tcc" "k_propagate|k_draw" tools/dense_bench.py > $OUT/pmc_dense.log 2>&1 || echo "pmc_dense failed"
# (the raw counter files of the model / dense passes are too large to travel back: their one derived figure is computed here)
cd $R && python3 tools/collect_valu_issue.py $OUT/pmc_models $OUT/pmc_dense > $OUT/valu_issue.json 2> $OUT/valu_issue.err || echo "valu_issue failed"
fi
# what travels back is limited to 64 MiB: the raw per-launch traces and the stamp dumps have been summarised above
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*.npy' -delete; find $OUT -name '*_agent_info.csv' -delete
rm -rf $OUT/pmc_sharded/pmc_* $OUT/pmc_sharded/trace $OUT/pmc_models/pmc_* $OUT/pmc_models/trace $OUT/pmc_dense/pmc_* $OUT/pmc_dense/trace
echo "bench done"
