import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.default_context(0)
P, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 10000)
s = syn.ba_scene(P=P, M=M, seed=7)
for _ in range(3):
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    t0 = time.perf_counter(); slam.bundle_adjustment_(cache, s["cam"]); dt = time.perf_counter() - t0
    print(f"BA P={P} O={s['O']} wall ms {dt*1e3:.2f} device ms {cache.stats['device_ms']:.3f} iters {cache.stats['iters_pass1']}+{cache.stats['iters_pass2']}")
