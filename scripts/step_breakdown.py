"""Host-side wall-time breakdown of one bench step (scratch tool)."""
import sys, time, collections
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
fast = "fast" in sys.argv
H, W = syn.SHAPES["kitti05"]
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream("kitti05", 8, seed=0, disparity=12.4)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in left]
rd = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in right]
ctxs = [slam.Context(0) for _ in range(3)]
be = bench.GpuBackend(slam, ctxs[0], ctxs[1], ctxs[2], H, W, ld, rd, params, ex, fast=fast)
T = collections.defaultdict(float); N = collections.defaultdict(int)
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] += time.perf_counter() - t0; N[name] += 1; return r
    setattr(obj, name, g)
for n in ("begin_frame", "match", "detect"): wrap(be, n)
st = bench.Stream(be, flows, 12.4, seed=0)
seq = bench.frame_sequence(400)
be.prime(seq[0])
for i in range(30): st.step(seq[i], seq[i+1], seq[i+2])
T.clear(); N.clear()
be.drain(); t0 = time.perf_counter()
K = 200
for i in range(30, 30 + K): st.step(seq[i], seq[i+1], seq[i+2])
be.drain(); tot = time.perf_counter() - t0
print(f"fast={fast} total {tot/K*1e3:.3f} ms/step")
acc = 0
for k in T: print(f"  {k:12s} {T[k]/K*1e3:.3f} ms/step  ({N[k]} calls, {T[k]/N[k]*1e6:.0f} us/call)"); acc += T[k]
print(f"  python glue  {(tot-acc)/K*1e3:.3f} ms/step")
