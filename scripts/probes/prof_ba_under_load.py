"""Local BA (50-KF window) alone vs while the lock-stepped front-end loop runs on the same GPU (the estimator thread beside the front-end,
estimator.jl:308-355): python scripts/prof_ba_under_load.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
import bench
dev = torch.device("cuda", 0)
s = syn.ba_scene(P=50, M=10000, seed=7)
ctx_ba = slam.Context(0)
def solve():
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(cache, s["cam"], ctx=ctx_ba)
    it = cache.stats["iters_pass1"] + cache.stats["iters_pass2"]
    return cache.stats["device_ms"] / it, cache.stats["ssr_final"]
solve()
alone = [solve()[0] for _ in range(10)]
print(f"alone: {np.median(alone) * 1e3:.1f} us per LM iteration (min {min(alone) * 1e3:.1f})")
wl = bench.make_workload(slam, syn, "kitti05_1000", seed=0, streams=64)
stop = [False]; under = []
def ba_thread():
    while not stop[0]:
        under.append(solve()[0]); time.sleep(0.002)
t = threading.Thread(target=ba_thread); t.start()
r = bench.run_lockstep_kpset(slam, torch, 0, wl, 20, 2, 1, None, dev, "host_u8")
stop[0] = True; t.join()
print(f"under load: {np.median(under) * 1e3:.1f} us per LM iteration (min {min(under) * 1e3:.1f}, n {len(under)}); front-end beside it: {r['value']:.0f} frames/s")
