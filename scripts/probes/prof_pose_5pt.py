"""slam_kpset_compute_pose_5pt (five-point RANSAC on device-resident lists with key-frame observations, 32 scenes), wall clock:
python scripts/prof_pose_5pt.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.Context(0)
SB = 32
fss = [syn.five_point_scene(n=1000, seed=40 + z, noise_px=0.4, outlier_frac=0.25, iters=128) for z in range(SB)]
cam = syn.KITTI_CAM
ks = slam.KeypointSet(SB, 1024, ctx=ctx)
for z, f in enumerate(fss):
    ks.upload(z, f["px2"][:, ::-1], np.zeros(len(f["px2"]), bool))
    ks.upload_keyframe(z, f["px1"][:, ::-1], np.ones(len(f["px1"]), bool))
sp = slam.stream_params(SB, Tcw=np.eye(4), cam=cam)
r = ks.compute_pose_5pt(sp, iters=128, seed=1, ctx=ctx)
t0 = time.perf_counter()
for i in range(20):
    r = ks.compute_pose_5pt(sp, iters=128, seed=2 + i, ctx=ctx)
print("kpset compute_pose_5pt ms", (time.perf_counter() - t0) / 20 * 1e3, "accepted", int(r[1].sum()), "pairs", r[4].mean())
t0 = time.perf_counter()
for i in range(20):
    rg = ks.compute_pose_5pt(sp, min_parallax=1e9, iters=128, seed=2 + i, ctx=ctx)
print("gated (no parallax) ms", (time.perf_counter() - t0) / 20 * 1e3, "accepted", int(rg[1].sum()))
