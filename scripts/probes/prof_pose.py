"""P3P / five-point RANSAC entry points only (for rocprofv3 --kernel-trace): python3 scripts/prof_pose.py [n] [tuples]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
it5 = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = slam.Context(0)
ps = syn.p3p_scene(n=n, seed=3, noise_px=0.4, outlier_frac=0.25, iters=256)
fs = syn.five_point_scene(n=n, seed=3, noise_px=0.4, outlier_frac=0.25, iters=it5)
for name, fn in (("p3p_ransac", lambda: slam.p3p_ransac(ps["pts3d"], ps["px_xy"], ps["pdn"], ps["K"], threshold=3.0, samples=ps["samples"], ctx=ctx)[0]),
                 ("five_point_ransac", lambda: slam.five_point_ransac(fs["px1"], fs["px2"], fs["pd1"], fs["pd2"], fs["K"], fs["K"], 3.0, samples=fs["samples"], ctx=ctx)[0])):
    fn()
    t0 = time.perf_counter()
    for _ in range(10):
        cnt = fn()
    print(name, "inliers", cnt, "wall ms/call", round((time.perf_counter() - t0) / 10 * 1e3, 3))
