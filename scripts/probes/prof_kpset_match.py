#!/usr/bin/env python3
"""k_kpset_match alone on the GPU: ns per keypoint of slam_kpset_flow_match at S streams (exact or tolerance-mode pyramids).
    python scripts/probes/prof_kpset_match.py [S] [exact|tol] [reps]      (env: SLAMHIP_LK_GRID, SLAMHIP_NO_TOL_LK)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tol = len(sys.argv) > 2 and sys.argv[2] == "tol"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H, W = syn.SHAPES['kitti05']
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream('kitti05', 4, seed=0, disparity=12.4)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(np.round(im * 255).astype(np.uint8).T)).to(dev) for im in left]
torch.cuda.synchronize()
ctx = slam.Context(0)
pb = [slam.PyramidBatch((H, W), levels=3, S=S, ctx=ctx) for _ in range(2)]
for k in range(2):
    pb[k].update_([ld[(s + k) % len(ld)].data_ptr() for s in range(S)], sync=True, ctx=ctx, u8=True, fast=tol)
base = [slam.detect(ex, pb[0].pyramids[s], np.zeros((0, 2)), ctx=ctx).astype(np.float64) for s in range(min(S, 4))]
rng = np.random.default_rng(0)
cap = 1400
ks = slam.KeypointSet(S, cap, ctx=ctx)
shift = np.array([np.array(flows[(s + 1) % len(ld)]) - np.array(flows[s % len(ld)]) for s in range(S)]) + rng.normal(0, 0.4, (S, 2))
sp = slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=shift)
tot_ms = 0.0; tot_n = 0
for r in range(reps + 1):
    n_in = 0
    for s in range(S):
        k = base[s % len(base)]
        ks.upload(s, k, np.arange(len(k)) % 5 != 0, ctx=ctx); n_in += len(k)
    ctx.prof_enable(True); ctx.prof_reset()
    ks.flow_match(pb[0], pb[1], params, sp, prior=2, n_bound=n_in, ctx=ctx)
    n_out = int(ks.counts(ctx=ctx).sum())
    ms, cnt = ctx.prof_get("fb_track"); ctx.prof_enable(False)
    if r > 0:
        tot_ms += ms; tot_n += n_in
print(f"S={S} tol={tol}: {tot_n // reps} keypoints per launch, {tot_ms / reps:.3f} ms per launch, {tot_ms * 1e6 / tot_n:.2f} ns per keypoint, kept {n_out / n_in:.3f}")
