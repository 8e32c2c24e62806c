import json, sys, os
sys.path.insert(0, os.getcwd())
import bench, modppl_amd
ys = bench.lgssm_observations(80)
for rep in range(3):
    r = bench.reference_shaped_loop(modppl_amd.lgssm_model(*bench.LGSSM_PARAMS), 1 << 20, ys, 50, 5)
    print(json.dumps({k: r[k] for k in ("us_per_step", "kernel_launches_per_step", "kernel_avg_us", "sum_of_log_total_weights")}))
