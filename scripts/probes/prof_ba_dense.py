"""The general solver path alone (26 poses, 24 observers per point: no banded order; for rocprofv3 --kernel-trace --stats)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.default_context(0)
s = syn.ba_scene(P=26, M=5200, seed=9, obs_per_point=24)
for _ in range(3):
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(cache, s["cam"])
print("device ms", cache.stats["device_ms"], "iters", cache.stats["iters_pass1"] + cache.stats["iters_pass2"])
