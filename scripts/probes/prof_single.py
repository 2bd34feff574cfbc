"""Host-side cost of the single-stream calls: python scripts/prof_single.py
 - slam_pyr_update (graph replay, sync=False): host time per enqueue, device time per build (events), builds back to back
 - the same with two builds in flight on two contexts
 - slam_flow_match (synchronous): wall time per call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
H, W = syn.SHAPES["kitti05"]
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream("kitti05", 2, seed=0)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in left]
torch.cuda.synchronize()
c = [slam.Context(0) for _ in range(3)]
pyr = [slam.LKPyramid(shape=(H, W), levels=3, ctx=c[0]) for _ in range(4)]
for p in pyr:
    slam.update_(p, None, device_ptr=ld[0].data_ptr(), sync=True, ctx=c[0])
slam.update_(pyr[1], None, device_ptr=ld[1].data_ptr(), sync=True, ctx=c[1])
N = 200
def builds(ctxs):
    for cc in ctxs: cc.synchronize()
    t0 = time.perf_counter(); th = 0.0
    for i in range(N):
        t1 = time.perf_counter()
        slam.update_(pyr[i % 4], None, device_ptr=ld[i % 2].data_ptr(), sync=False, ctx=ctxs[i % len(ctxs)])
        th += time.perf_counter() - t1
    for cc in ctxs: cc.synchronize()
    return (time.perf_counter() - t0) / N * 1e6, th / N * 1e6
for k in (1, 2, 3):
    w, h = builds(c[:k])
    print(f"{k} build stream(s): {w:.1f} us per build wall, host enqueue {h:.1f} us")
kp = slam.detect(ex, pyr[0], np.zeros((0, 2)), ctx=c[0]).astype(float)
is3 = np.arange(len(kp)) % 10 != 0
proj = kp + np.array(flows[1])
slam.update_(pyr[0], None, device_ptr=ld[0].data_ptr(), sync=True, ctx=c[0]); slam.update_(pyr[1], None, device_ptr=ld[1].data_ptr(), sync=True, ctx=c[0])
t0 = time.perf_counter()
for _ in range(N):
    new, st = slam.optical_flow_matching(pyr[0], pyr[1], kp, is3, proj, params, ctx=c[0])
print(f"optical_flow_matching ({len(kp)} kpts): {(time.perf_counter() - t0) / N * 1e6:.1f} us per call wall; ok {st.mean():.3f}")
c[0].prof_enable(True); c[0].prof_reset()
for _ in range(50):
    slam.optical_flow_matching(pyr[0], pyr[1], kp, is3, proj, params, ctx=c[0])
ms, n = c[0].prof_get("fb_track"); print(f"  kernel {ms / n * 1e3:.1f} us")
t0 = time.perf_counter()
for _ in range(N):
    slam.detect(ex, pyr[0], kp, ctx=c[0])
print(f"detect: {(time.perf_counter() - t0) / N * 1e6:.1f} us per call wall")
# builds while tracking (the bench's single-stream loop without key-frames)
for k in (1, 2):
    for cc in c: cc.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        slam.update_(pyr[2 + i % 2], None, device_ptr=ld[i % 2].data_ptr(), sync=False, ctx=c[1 + (i % k)])
        slam.optical_flow_matching(pyr[0], pyr[1], kp, is3, proj, params, ctx=c[0])
    for cc in c: cc.synchronize()
    print(f"build (on {k} stream(s)) + synchronous match per frame: {(time.perf_counter() - t0) / N * 1e6:.1f} us")
