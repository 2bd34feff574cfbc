#!/usr/bin/env python3
"""slam_local_ba_batch: wall / device time of S reference-shaped windows (5 free + 20 constant key-frames) and of S x P20.
    python scripts/probes/ba_batch_time.py [S]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import slam_jl_amd as slam  # noqa: E402
from slam_jl_amd import synthetic as syn  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = slam.default_context(0)
ONLY = sys.argv[2] if len(sys.argv) > 2 else None
for name, mk in (("P5_free_20_const", lambda z: syn.ba_scene(P=25, M=800, seed=100 + z, n_const=20)), ("P20", lambda z: syn.ba_scene(P=20, M=4000, seed=300 + z))):
    if ONLY and name != ONLY:
        continue
    base = [mk(z) for z in range(8)]
    sc = [base[z % 8] for z in range(S)]
    def caches():
        return [slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"]) for s in sc]
    c = caches(); slam.bundle_adjustment_batch_(c, sc[0]["cam"])
    best = None
    for _ in range(5):
        c = caches()
        t0 = time.perf_counter(); slam.bundle_adjustment_batch_(c, sc[0]["cam"]); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    it = np.mean([x.stats["iters_pass1"] + x.stats["iters_pass2"] for x in c])
    one = slam.LocalBACache(sc[0]["theta0"].copy(), sc[0]["theta_const"], sc[0]["pixels_yx"], sc[0]["pose_ids"], sc[0]["point_ids"])
    slam.bundle_adjustment_(one, sc[0]["cam"])
    t0 = time.perf_counter(); slam.bundle_adjustment_(one, sc[0]["cam"]); d1 = time.perf_counter() - t0
    print(f"{name}: S = {S}: wall {best * 1e3:.2f} ms (python marshalling included), device {c[0].stats['device_ms']:.3f} ms, {S / best:.0f} windows/s, "
          f"mean LM iterations {it:.1f}; one window alone: wall {d1 * 1e3:.2f} ms, device {one.stats['device_ms']:.3f} ms", flush=True)
