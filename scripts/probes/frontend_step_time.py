import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
H, W = syn.SHAPES["kitti05"]
L, R, flows = syn.stereo_stream("kitti05", 8, seed=0, step=(1.3, -2.1), disparity=12.4)
u8 = lambda im: np.ascontiguousarray(np.round(np.clip(im, 0, 1) * 255).astype(np.uint8).T)
L8 = [u8(x) for x in L]; R8 = [u8(x) for x in R]; fl = np.array(flows)
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W); ex = slam.Extractor.from_params(params, cam); camt = tuple(syn.KITTI_CAM)
T21 = np.eye(4); T21[0, 3] = -0.54
tri = slam.FrontEnd.tri_params(camt, camt, T21, np.eye(4)); sps = slam.stream_params(1, cam=camt, shift_yx=(0.0, -12.4))
seq = [0, 1, 2, 3, 4, 5, 6, 7, 6, 5, 4, 3, 2, 1] * 40
Lp = [torch.from_numpy(x).pin_memory() for x in L8]; Rp = [torch.from_numpy(x).pin_memory() for x in R8]
import os
for fast in (False, True):
    for la in ((True,) if os.environ.get('FE_LA_ONLY') else (False, True)):
      for raw in ((True,) if os.environ.get('FE_LA_ONLY') else (False, True)):
        fe = slam.FrontEnd((H, W), params, ex, fast=fast, lookahead=la)
        pr = [slam.stream_params(1, cam=camt, shift_yx=(fl[seq[t]] - fl[seq[t - 1]]) if t > 0 else (0, 0)) for t in range(len(seq))]
        def call(t):
            due = t - 1 if la else t
            if raw:
                return fe.step_ptr(Lp[seq[t]].data_ptr(), Rp[seq[t]].data_ptr() if t % 5 == 0 else 0, pr[max(due, 0)].ctypes.data, 2, sps.ctypes.data, 2, tri.ctypes.data)
            return fe.step(L8[seq[t]], R8[seq[t]] if t % 5 == 0 else None, params=pr[max(due, 0)], prior=2, stereo_params=sps, stereo_prior=2, tri=tri)
        for t in range(30): call(t)
        t0 = time.perf_counter()
        for t in range(30, 330): fr, cnt = call(t)
        dt = time.perf_counter() - t0
        print(f"fast={fast} lookahead={la} pinned-frames+raw-pointers={raw}: {300 / dt:.0f} frames/s ({dt / 300 * 1e6:.0f} us per call), list {cnt}", flush=True)
        fe.close()
