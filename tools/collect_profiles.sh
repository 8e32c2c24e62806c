#!/bin/bash
# Build side, after `gpurun -- bash tools/profile_round.sh rNN_prof`: copy the summaries to be judged from gpurun_out/ (scratch)
# into profiles/rNN (tracked).    tools/collect_profiles.sh r03
set -eu
R=${1:-r05}
SRC=gpurun_out/${R}_prof
DST=profiles/$R
mkdir -p $DST
# the run must be of THIS tree (traffic.json / c4_flops.json are stamped with the local content hash)
if [ -f $SRC/source_hash.txt ]; then
  HERE=$(python3 -c "from modppl_amd import build as B; print(B.source_hash())")
  [ "$(cat $SRC/source_hash.txt)" = "$HERE" ] || { echo "profile run is of another build ($(cut -c1-8 $SRC/source_hash.txt) vs $(echo $HERE | cut -c1-8)): not collected"; exit 1; }
else
  echo "warning: $SRC/source_hash.txt missing (a run that did not travel back?)"; exit 1
fi
python3 tools/collect_traffic.py $SRC > $DST/traffic.json
python3 tools/pmc_summary.py $SRC > $DST/counters_summary.txt
[ -f $SRC/valu_issue.json ] && cp $SRC/valu_issue.json $DST/valu_issue.json
for f in stamp_report.txt stamp_report_tile_kernel.txt k1_scaling.txt k1_forms_ab.txt reference_shaped_loop.jsonl reference_shaped_loop_cpp.txt stamps.json bench_n1.json bench_n1_k20.json bench_forced_sharded.json bench_forced_sharded_rccl.json bench_forced_sharded_split.json route_scale.txt table_stamps.txt table_stamps_two_calls.txt model_bench.jsonl mh_functor_vs_handwritten.json sync_probe.jsonl; do
  [ -f $SRC/$f ] && cp $SRC/$f $DST/$f
done
cp $SRC/trace/bench_kernel_stats.csv $DST/kernel_stats.csv
cp $SRC/trace_sharded/bench_kernel_stats.csv $DST/kernel_stats_forced_sharded.csv
[ -f $SRC/trace_sharded_split/bench_kernel_stats.csv ] && cp $SRC/trace_sharded_split/bench_kernel_stats.csv $DST/kernel_stats_forced_sharded_split.csv
for i in 1 2 3 4 5 6 7; do cp $SRC/pmc$i/p_counter_collection.csv $DST/pmc${i}_counter_collection.csv; done
for d in pmc_models pmc_mh pmc_dense pmc_sharded; do [ -f $SRC/$d/summary.txt ] && cp $SRC/$d/summary.txt $DST/${d}_summary.txt; done
python3 tools/collect_c4_flops.py $SRC/pmc_mh 30 > $DST/c4_flops.json || echo "no C4 flop summary"

echo "collected into $DST"
