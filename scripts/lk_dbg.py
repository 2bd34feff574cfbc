"""scratch: per-point comparison of slam_fb_track with the oracle (half-wave kernel bring-up)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
from oracle import oracle as orc
H, W = 120, 160
L, R, flows = syn.stereo_stream((H, W), 2, seed=3)
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sub = len(sys.argv) > 3
kp = orc.detect(L[0], np.zeros((0, 2)), max_points=1000).astype(float)
if sub:
    kp = np.clip(kp + np.random.default_rng(1).uniform(0, 0.99, kp.shape), 1, [H, W])
kp = kp[:n]
a = slam.LKPyramid(shape=(H, W), levels=3); slam.update_(a, L[0]); b = slam.LKPyramid(shape=(H, W), levels=3); slam.update_(b, L[1])
ra, rb = orc.pyr_build(L[0], 3, 1.0, 1), orc.pyr_build(L[1], 3, 1.0, 1)
out, st = slam.fb_tracking_(a, b, kp, window_size=9, pyramid_levels=levels, max_distance=1.0)
ro, rs = orc.fb_tracking(ra, rb, kp, pyramid_levels=levels, sum_order=2)
bad = 0
for i in range(len(kp)):
    d = np.abs(out[i] - ro[i]).max() if st[i] and rs[i] else float('nan')
    if st[i] != rs[i] or d > 1e-9:
        bad += 1
        if bad < 25: print(i, i % 2, kp[i], "gpu", bool(st[i]), "orc", bool(rs[i]), "diff", d)
print("n", len(kp), "bad", bad, "gpu ok", int(st.sum()), "orc ok", int(rs.sum()))
