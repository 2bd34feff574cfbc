"""S-stream batched flow match only (for rocprofv3 --pmc): python3 scripts/prof_flow.py [S] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
its = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H, W = syn.SHAPES['kitti05']
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream('kitti05', 4, seed=0, disparity=12.4)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in left]
torch.cuda.synchronize()
ctx = slam.Context(0)
pb = [slam.PyramidBatch((H, W), levels=params.pyramid_levels, S=S, ctx=ctx) for _ in range(2)]
for k in range(2):
    pb[k].update_([ld[(s + k) % len(ld)].data_ptr() for s in range(S)], sync=True, ctx=ctx)
kps, sids = [], []
for s in range(S):
    k = slam.detect(ex, pb[0].pyramids[s], np.zeros((0, 2)), ctx=ctx).astype(np.float64)
    kps.append(k); sids.append(np.full(len(k), s, np.int32))
kp = np.concatenate(kps); sid = np.concatenate(sids)
is3d = np.arange(len(kp)) % 10 != 0
fl = np.array([np.array(flows[(s + 1) % len(ld)]) - np.array(flows[s % len(ld)]) for s in range(S)])
proj = kp + fl[sid]
for _ in range(5):
    new, ok = slam.optical_flow_matching_batch(pb[0], pb[1], sid, kp, is3d, proj, params, iterations=its, ctx=ctx)
print(len(kp), ok.mean())
import ctypes as C, time
from slam_jl_amd import _lib as L
def raw(n, eig=1e-4, its=30, lv3=1, lv=params.pyramid_levels, reps=10):
    idx = np.ascontiguousarray(sid[:n]); px = np.ascontiguousarray(kp[:n]); i3 = np.ascontiguousarray(is3d[:n].astype(np.uint8)); pj = np.ascontiguousarray(proj[:n])
    out = np.empty((n, 2)); st = np.zeros(n, dtype=np.uint8)
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.check(ctx.lib.slam_flow_match_batch(ctx.h, pb[0].pyramids[0].h, pb[1].pyramids[0].h, S, L.ptr(idx, L.i32p), L.ptr(px), L.ptr(i3, L.u8p), L.ptr(pj), n,
                                                lv, lv3, params.window_size, its, eig, 1e-2, float(params.max_ktl_distance), L.ptr(out), L.ptr(st, L.u8p)))
    wall = (time.perf_counter() - t0) / reps * 1e6
    ms, cnt = ctx.prof_get("fb_track")
    ctx.prof_enable(False)
    return round(wall, 1), round(ms / cnt * 1e3, 1), float(st.mean())
perm = np.random.default_rng(0).permutation(len(kp))
print("n, (wall us, kernel us, ok)")
for n in (1155, 2310, 4620, 9240, 18480):
    print(n, raw(n))
print("eig_thr=1e30 (setup only)", raw(len(kp), eig=1e30))
print("iterations=1", raw(len(kp), its=1))
print("iterations=0", raw(len(kp), its=0))
print("levels 0/0", raw(len(kp), lv3=0, lv=0))
