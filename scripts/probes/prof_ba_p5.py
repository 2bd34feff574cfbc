"""The reference-shaped window alone (25 poses, 20 constant; for rocprofv3 --kernel-trace --stats): python scripts/prof_ba_p5.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.default_context(0)
s = syn.ba_scene(P=25, M=800, seed=5, n_const=20)
for _ in range(4):
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(cache, s["cam"])
print("device ms", cache.stats["device_ms"], "iters", cache.stats["iters_pass1"] + cache.stats["iters_pass2"])
