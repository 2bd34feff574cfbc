import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
for name, s in (("P5_free_20_const", syn.ba_scene(P=25, M=800, seed=5, n_const=20)), ("P20", syn.ba_scene(P=20, M=4000, seed=6)), ("P50", syn.ba_scene(P=50, M=10000, seed=7)), ("P100", syn.ba_scene(P=100, M=40000, seed=8))):
    for _ in range(2):
        c = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"]); slam.bundle_adjustment_(c, s["cam"])
    ts = []
    for _ in range(10):
        c = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        t0 = time.perf_counter(); slam.bundle_adjustment_(c, s["cam"]); ts.append(time.perf_counter() - t0)
    print(name, "O", s["O"], "wall ms", round(min(ts) * 1e3, 3), "device ms", round(c.stats["device_ms"], 3), "iters", c.stats["iters_pass1"], c.stats["iters_pass2"])
