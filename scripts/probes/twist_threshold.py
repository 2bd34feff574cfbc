"""ms per LM iteration around the twisted solve's threshold (hb = 9): python scripts/twist_threshold.py   [SLAMHIP_TWIST_MIN=17 for the lower one]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.Context(0)
row = []
for P in (17, 18, 19, 20, 21, 22, 24):
    s = syn.ba_scene(P=P, M=200 * P, seed=6)
    best = 1e9
    for _ in range(4):
        cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        slam.bundle_adjustment_(cache, s["cam"], ctx=ctx)
        best = min(best, cache.stats["device_ms"] / (cache.stats["iters_pass1"] + cache.stats["iters_pass2"]))
    row.append(f"{P}:{best * 1e3:.1f}")
print("us per LM iteration by poses:", " ".join(row))
