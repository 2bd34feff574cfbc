"""Quick per-stage timing of the seams on one KITTI-sized stereo pair (scratch tool)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn

H, W = 370, 1226
L, R, flows = syn.stereo_stream((H, W), 3, seed=0)
ctx = slam.default_context(0)
e = slam.Extractor(1000, 17, (-(-H // 35), -(-W // 35)), 35)
prev = slam.LKPyramid(shape=(H, W), levels=3); cur = slam.LKPyramid(shape=(H, W), levels=3)
slam.update_(prev, L[0]); slam.update_(cur, L[1])
kp = slam.detect(e, prev, np.zeros((0, 2))).astype(float)
print("kp", len(kp))
def t(f, n=20):
    f(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("pyr_update host img  ms", t(lambda: slam.update_(cur, L[1])))
print("detect_pyr           ms", t(lambda: slam.detect(e, prev, kp[:500])))
print("fb_track 3 levels    ms", t(lambda: slam.fb_tracking_(prev, cur, kp, window_size=9, pyramid_levels=3, max_distance=1.0)))
print("fb_track 1 level     ms", t(lambda: slam.fb_tracking_(prev, cur, kp, displacement=np.tile(np.array(flows[1])/2, (len(kp),1)), window_size=9, pyramid_levels=1, max_distance=1.0)))
for P, M in ((5, 800), (20, 4000), (50, 10000)):
    s = syn.ba_scene(P=P, M=M, seed=1)
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    t0 = time.perf_counter(); slam.bundle_adjustment_(cache, s["cam"]); dt = time.perf_counter() - t0
    print(f"BA P={P} O={s['O']} wall ms {dt*1e3:.2f} device ms {cache.stats['device_ms']:.3f} iters {cache.stats['iters_pass1']}+{cache.stats['iters_pass2']} ssr {cache.stats['ssr_final']:.2f}")
