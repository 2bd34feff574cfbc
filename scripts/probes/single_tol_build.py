#!/usr/bin/env python3
"""ONE image through the single-image builds, for a kernel trace: python scripts/probes/single_tol_build.py [exact|tol] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
tol = len(sys.argv) > 1 and sys.argv[1] == "tol"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, W = syn.SHAPES['kitti05']
left, right, flows = syn.stereo_stream('kitti05', 2, seed=0)
dev = torch.device("cuda", 0)
img = torch.from_numpy(np.ascontiguousarray(left[0].T)).to(dev); torch.cuda.synchronize()
ctx = slam.Context(0)
p = slam.LKPyramid(shape=(H, W), levels=3, ctx=ctx)
import ctypes as C
mode = 3 if tol else 1
for _ in range(3):
    ctx.check(ctx.lib.slam_pyr_update_dev(ctx.h, p.h, C.c_void_p(img.data_ptr()), mode, 1.0, 1))
t0 = time.perf_counter()
for _ in range(reps):
    ctx.check(ctx.lib.slam_pyr_update_dev(ctx.h, p.h, C.c_void_p(img.data_ptr()), mode, 1.0, 0))
ctx.synchronize()
print(f"tol={tol}: {(time.perf_counter() - t0) / reps * 1e6:.1f} us per build (back to back)")
time.sleep(0.01)
ctx.check(ctx.lib.slam_pyr_update_dev(ctx.h, p.h, C.c_void_p(img.data_ptr()), mode, 1.0, 1))      # the traced one, alone
