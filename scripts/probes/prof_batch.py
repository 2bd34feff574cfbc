"""Breakdown of the lock-step batched step: python scripts/prof_batch.py [S] [fast]
(run under rocprofv3 --kernel-trace --stats for per-kernel numbers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fast = len(sys.argv) > 2 and sys.argv[2] == "fast"
H, W = syn.SHAPES["kitti05"]
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream("kitti05", 8, seed=0, disparity=12.4)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in left]
rd = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in right]
torch.cuda.synchronize()
ctx = slam.Context(0)
pb = [slam.PyramidBatch((H, W), levels=params.pyramid_levels, S=S, ctx=ctx) for _ in range(2)]
ptrs = [[ld[(s + k) % len(ld)].data_ptr() for s in range(S)] for k in range(2)]
for k in range(2):
    pb[k].update_(ptrs[k], sync=True, fast=fast, ctx=ctx)
def t(fn, n=50):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("S", S, "fast", fast)
print("pyr batch update us", t(lambda: pb[1].update_(ptrs[1], sync=False, fast=fast, ctx=ctx)))
kps = []; sids = []
for s in range(S):
    k = slam.detect(ex, pb[0].pyramids[s], np.zeros((0, 2)), ctx=ctx).astype(np.float64)
    kps.append(k); sids.append(np.full(len(k), s, np.int32))
print("detect per stream us", t(lambda: slam.detect(ex, pb[0].pyramids[0], kps[0][:800], ctx=ctx)))
kp = np.concatenate(kps); sid = np.concatenate(sids)
is3d = np.arange(len(kp)) % 10 != 0
fl = np.array([np.array(flows[(s + 1) % len(ld)]) - np.array(flows[s % len(ld)]) for s in range(S)])
proj = kp + fl[sid]
print("points", len(kp))
print("flow batch us", t(lambda: slam.optical_flow_matching_batch(pb[0], pb[1], sid, kp, is3d, proj, params, ctx=ctx)))
new, ok = slam.optical_flow_matching_batch(pb[0], pb[1], sid, kp, is3d, proj, params, ctx=ctx)
print("ok frac", ok.mean())
rng = np.random.default_rng(0)
def glue():
    p = kp + fl[sid] + rng.normal(0, 0.5, kp.shape)
    a, b, c = kp[ok], is3d[ok], sid[ok]
print("numpy glue us", t(glue))
for its in (1, 2, 3, 5, 10, 20, 30):
    print("iterations", its, "flow batch us", t(lambda: slam.optical_flow_matching_batch(pb[0], pb[1], sid, kp, is3d, proj, params, iterations=its, ctx=ctx), n=10))
none3d = np.zeros(len(kp), dtype=bool)
print("all-2D (3 levels, no prior) us", t(lambda: slam.optical_flow_matching_batch(pb[0], pb[1], sid, kp, none3d, proj, params, ctx=ctx), n=10))
all3d = np.ones(len(kp), dtype=bool)
print("all-3D us", t(lambda: slam.optical_flow_matching_batch(pb[0], pb[1], sid, kp, all3d, proj, params, ctx=ctx), n=10))
