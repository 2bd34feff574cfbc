"""one single-window BA solve of the general-path window of bench.py (P26_dense: every point seen by 24 of 26 key-frames), for a kernel trace"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.default_context(0)
s = syn.ba_scene(P=26, M=5200, seed=9, obs_per_point=24)
for _ in range(3):
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    t0 = time.perf_counter(); slam.bundle_adjustment_(cache, s["cam"]); dt = time.perf_counter() - t0
    print(f"BA P26_dense O={s['O']} wall ms {dt*1e3:.2f} device ms {cache.stats['device_ms']:.3f} iters {cache.stats['iters_pass1']}+{cache.stats['iters_pass2']}")
