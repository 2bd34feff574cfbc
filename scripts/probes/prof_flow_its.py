"""Batched LK with a fixed iteration cap only (for rocprofv3 --pmc: instruction counts of set-up vs iterations):
python3 scripts/prof_flow_its.py S its [noise_px]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
S = int(sys.argv[1]); its = int(sys.argv[2]); noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
H, W = syn.SHAPES['kitti05']
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream('kitti05', 4, seed=0, disparity=12.4)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in left]
torch.cuda.synchronize()
ctx = slam.Context(0)
pb = [slam.PyramidBatch((H, W), levels=3, S=S, ctx=ctx) for _ in range(2)]
for k in range(2):
    pb[k].update_([ld[(s + k) % len(ld)].data_ptr() for s in range(S)], sync=True, ctx=ctx)
kps, sids = [], []
for s in range(S):
    k = slam.detect(ex, pb[0].pyramids[s], np.zeros((0, 2)), ctx=ctx).astype(np.float64)
    kps.append(k); sids.append(np.full(len(k), s, np.int32))
kp = np.concatenate(kps); sid = np.concatenate(sids)
is3d = np.arange(len(kp)) % 10 != 0
fl = np.array([np.array(flows[(s + 1) % len(ld)]) - np.array(flows[s % len(ld)]) for s in range(S)])
proj = kp + fl[sid] + np.random.default_rng(0).normal(0, noise, kp.shape)
for _ in range(6):
    new, ok = slam.optical_flow_matching_batch(pb[0], pb[1], sid, kp, is3d, proj, params, iterations=its, ctx=ctx)
print(len(kp), its, ok.mean())
