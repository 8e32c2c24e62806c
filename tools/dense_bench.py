import sys, json, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
r = bench.sub_benches(20, 3, ("c5",))
print(json.dumps({k: {a: b for a, b in v.items() if a in ("us_per_step", "particle_steps_per_s", "step_hbm_frac", "log_ml")} for k, v in r.items()}))
