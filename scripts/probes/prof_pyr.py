"""Timing of the pyramid build variants (scratch tool). usage: prof_pyr.py [fast] [H W]"""
import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
fast = "fast" in sys.argv
H, W = 370, 1226
L, R, flows = syn.stereo_stream((H, W), 2, seed=0)
ctx = slam.default_context(0)
cur = slam.LKPyramid(shape=(H, W), levels=3)
dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in L]
torch.cuda.synchronize()
for i in range(5): slam.update_(cur, None, device_ptr=dev[i % 2].data_ptr(), fast=fast)
ctx.synchronize(); t0 = time.perf_counter()
N = 200
for i in range(N): slam.update_(cur, None, device_ptr=dev[i % 2].data_ptr(), sync=False, fast=fast)
ctx.synchronize(); dt = (time.perf_counter() - t0) / N
print(f"pyramid update (device image, {'fast' if fast else 'exact'}, graph={'no' if os.environ.get('SLAMHIP_NO_GRAPH') else 'yes'}): {dt*1e6:.1f} us")
