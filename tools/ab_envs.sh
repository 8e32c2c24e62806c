#!/bin/bash
# same-box comparison of several values of one environment switch on the bench: tools/ab_envs.sh VAR reps v1 v2 ...  (round-robin)
# (the switches exist in the DIAGNOSTICS build of the library only, csrc/mp_diag.h: python -m modppl_amd.build diag)
VAR=$1; REPS=$2; shift; shift
export MODPPL_HIP_LIB=$PWD/modppl_amd/csrc/libmodppl_hip_diag.so
for i in $(seq $REPS); do
  for V in "$@"; do
    env $VAR=$V timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline --no-systematic-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$VAR=$V', round(d['ms_per_step'] * 1e3, 2), 'us', {k: round(v, 2) for k, v in (d.get('kernel_avg_us') or {}).items()})"
  done
done
