"""Batched detect alone: python scripts/prof_detect.py [S]  (with / without current keypoints = with / without the avoidance mask)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch

import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = syn.SHAPES["kitti05"]
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream("kitti05", 2, seed=0, disparity=12.4)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in left]
torch.cuda.synchronize()
ctx = slam.Context(0)
pb = slam.PyramidBatch((H, W), levels=params.pyramid_levels, S=S, ctx=ctx)
pb.update_([ld[s % 2].data_ptr() for s in range(S)], sync=True, ctx=ctx)
kp0, sid0 = slam.detect_batch(ex, pb, np.zeros((0, 2)), np.zeros(0, dtype=np.int32), ctx=ctx)
kp0 = kp0.astype(np.float64)
def t(fn, n=10):
    """device time of the call's kernels (hipEvent span "detect" of the library), us"""
    fn(); ctx.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(n): fn()
    ctx.synchronize()
    ms, cnt = ctx.prof_get("detect"); ctx.prof_enable(False)
    return ms / max(cnt, 1) * 1e3
print("S", S, "keypoints", len(kp0))
print("no current points (no mask): us", round(t(lambda: slam.detect_batch(ex, pb, np.zeros((0, 2)), np.zeros(0, dtype=np.int32), ctx=ctx))))
keep = np.arange(len(kp0)) % 100 < 85
print("85 % current points (mask, k=1): us", round(t(lambda: slam.detect_batch(ex, pb, kp0[keep], sid0[keep], ctx=ctx))))
keep = np.arange(len(kp0)) % 100 < 30
print("30 % current points (mask, k=2): us", round(t(lambda: slam.detect_batch(ex, pb, kp0[keep], sid0[keep], ctx=ctx))))
keep = np.arange(len(kp0)) % 100 < 85
print("85 % current points, sigma_mask = 0 (no blur): us", round(t(lambda: slam.detect_batch(ex, pb, kp0[keep], sid0[keep], sigma_mask=0.0, ctx=ctx))))
far = kp0[keep] * 0 + np.array([-1000.0, -1000.0])
print("85 % current points far outside (scan only, empty mask): us", round(t(lambda: slam.detect_batch(ex, pb, far, sid0[keep], ctx=ctx))))
