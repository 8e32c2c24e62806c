for i in 1 2; do
for e in "MP_FUSED_DRAWS=1" "MP_FUSED_DRAWS=0" "MP_FUSED_DRAWS=0 MP_K1_THREADS=512" "MP_FUSED_DRAWS=0 MP_K1_THREADS=256"; do
  echo -n "[$e] "
  env $e timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-sub-benches --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['ms_per_step'] * 1e3, 2), 'us', {k: round(v, 2) for k, v in (d.get('kernel_avg_us') or {}).items() if v})"
done; done
