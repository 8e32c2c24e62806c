#!/bin/bash
# same-box A/B of one environment switch on the bench: tools/ab_env.sh VAR valueA valueB [reps]   (alternating runs)
VAR=$1; A=$2; B=$3; REPS=${4:-3}
for i in $(seq $REPS); do
  for V in $A $B; do
    env $VAR=$V timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$VAR=$V', round(d['ms_per_step'] * 1e3, 2), 'us', {k: round(v, 2) for k, v in (d.get('kernel_avg_us') or {}).items()})"
  done
done
