"""Two half batches on two streams vs one batch: python scripts/prof_pyr_split.py [S] [parts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = 30
H, W = syn.SHAPES["kitti05"]
left, right, flows = syn.stereo_stream("kitti05", 8, seed=0, disparity=12.4)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in left]
torch.cuda.synchronize()
ctxs = [slam.Context(0) for _ in range(parts)]
Sp = S // parts
pbs = [slam.PyramidBatch((H, W), levels=3, S=Sp, ctx=c) for c in ctxs]
ptrs = [[ld[(s + k * Sp) % len(ld)].data_ptr() for s in range(Sp)] for k in range(parts)]
for _ in range(3):
    for k in range(parts):
        pbs[k].update_(ptrs[k], sync=False, ctx=ctxs[k])
for c in ctxs: c.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    for k in range(parts):
        pbs[k].update_(ptrs[k], sync=False, ctx=ctxs[k])
for c in ctxs: c.synchronize()
g = (time.perf_counter() - t0) / reps * 1e6
alg = S * bench.pyramid_bytes(H, W, 3)
print(f"S={S} in {parts} concurrent parts: {g:.1f} us per {S} images (frac {alg / g / 1e3 / 8000:.3f})")
