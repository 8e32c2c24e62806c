#!/bin/bash
# same-box A/B of two builds of the library on the forced-sharded (or plain) bench: tools/ab_bench.sh libA.so libB.so [reps]
# (each line: ms_per_step of A then of B, alternating so that clock drift hits both alike)
A=$1; B=$2; REPS=${3:-3}
for i in $(seq $REPS); do
  for L in $A $B; do
    MODPPL_HIP_LIB=$PWD/$L timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-sub-benches 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$L', round(d['ms_per_step'] * 1e3, 2), 'us')"
  done
done
