#!/bin/bash
# same-box comparison of several builds of the library on the bench: tools/ab_libs.sh reps lib1.so lib2.so ...  (round-robin)
REPS=$1; shift
for i in $(seq $REPS); do
  for L in "$@"; do
    MODPPL_HIP_LIB=$PWD/$L timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$L', round(d['ms_per_step'] * 1e3, 2), 'us', {k: round(v, 2) for k, v in (d.get('kernel_avg_us') or {}).items()})"
  done
done
