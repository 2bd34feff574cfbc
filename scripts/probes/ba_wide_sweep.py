"""Which (poses, observers per point) chains does the banded solver take without complaint?  python scripts/ba_wide_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.Context(0)
for P in (24, 30, 40, 45, 50, 60, 80, 100):
    row = []
    for k in (8, 10, 11, 12, 14, 16, 18, 19, 20, 21):
        if k >= P: continue
        s = syn.ba_scene(P=P, M=40 * P, seed=3, obs_per_point=k)
        cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        try:
            slam.bundle_adjustment_(cache, s["cam"], ctx=ctx)
            it = cache.stats["iters_pass1"] + cache.stats["iters_pass2"]
            row.append(f"{k}:{cache.stats['device_ms'] / max(it, 1):.3f}")
        except Exception as ex:
            row.append(f"{k}:FAIL")
    print(P, " ".join(row), flush=True)
