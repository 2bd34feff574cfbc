"""Host / device split of slam_local_ba per window: SLAMHIP_BA_HOSTTIME=1 python scripts/ba_hosttime.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
import bench
ctx = slam.Context(0)
for name, s in bench.ba_windows(syn).items():
    for rep in range(3):
        cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        t0 = time.perf_counter(); slam.bundle_adjustment_(cache, s["cam"], ctx=ctx); w = time.perf_counter() - t0
    print(f"{name}: wall {w * 1e3:.2f} ms, device {cache.stats['device_ms']:.2f} ms", flush=True)
