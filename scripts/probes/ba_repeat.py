"""Determinism / hand-over stress of the twisted banded solve: the same window solved N times must give bit-identical parameters."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for P, M in ((50, 10000), (31, 1500), (100, 8000)):
    s = syn.ba_scene(P=P, M=M, seed=3)
    ref = None; bad = 0
    for it in range(N):
        cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        slam.bundle_adjustment_(cache, s["cam"])
        key = (cache.theta.tobytes(), cache.outliers.tobytes(), cache.stats["ssr_final"])
        if ref is None: ref = key
        elif key != ref: bad += 1
    print(f"P={P} M={M}: {N} runs, {bad} differ from the first (ssr_final {ref[2]!r})")
