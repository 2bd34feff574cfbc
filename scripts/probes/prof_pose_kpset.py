"""slam_kpset_compute_pose (P3P RANSAC + PnP refinement on device-resident lists, 32 scenes) and the single-stream seams, wall clock:
python scripts/prof_pose_kpset.py   (A/B another build with SLAMHIP_LIB=...; under rocprofv3 --kernel-trace --stats for the kernel table)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.Context(0)
SB = 32
pss = [syn.p3p_scene(n=1000, seed=40 + z, noise_px=0.4, outlier_frac=0.25, iters=256) for z in range(SB)]
Kc = pss[0]["K"]; camp = (Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2])
ks = slam.KeypointSet(SB, 1024, ctx=ctx)
for z, q in enumerate(pss):
    ks.upload(z, q["px_xy"][:, ::-1], np.ones(len(q["pts3d"]), bool), q["pts3d"])
sp = slam.stream_params(SB, cam=camp)
ks.compute_pose(sp, iters=256, seed=1, ctx=ctx)
t0 = time.perf_counter()
for i in range(20):
    r = ks.compute_pose(sp, iters=256, seed=2 + i, ctx=ctx)
print("kpset compute_pose ms", (time.perf_counter() - t0) / 20 * 1e3, "accepted", int(r[1].sum()))
ps = pss[0]
def once():
    cnt, (KP, inl, err, Rt, bi) = slam.p3p_ransac(ps["pts3d"], ps["px_xy"], ps["pdn"], Kc, threshold=3.0, samples=ps["samples"], return_pose=True, ctx=ctx)
    T0 = np.eye(4); T0[:3] = Rt
    return slam.pnp_bundle_adjustment(camp, T0, ps["px_xy"][inl][:, ::-1], ps["pts3d"][inl], repr_eps=3.0, ctx=ctx)
once(); t0 = time.perf_counter()
for i in range(20): once()
print("single p3p+pnp ms", (time.perf_counter() - t0) / 20 * 1e3)
