#!/usr/bin/env python3
"""Host-side set-up of slam_local_ba_batch alone (no GPU needed): python scripts/probes/ba_host_time.py [S] [threads ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import _lib as L, synthetic as syn
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
thr = [int(x) for x in sys.argv[2:]] or [1, 4, 8]
lib = slam.load()
base = [syn.ba_scene(P=25, M=800, seed=100 + z, n_const=20) for z in range(8)]
b = slam.BABatch([slam.LocalBACache(base[z % 8]["theta0"].copy(), base[z % 8]["theta_const"], base[z % 8]["pixels_yx"], base[z % 8]["pose_ids"], base[z % 8]["point_ids"]) for z in range(S)], base[0]["cam"])
lib.slam_debug_ba_host_time.restype = C.c_int
out = np.zeros(3)
for t in thr:
    best = None
    for _ in range(5):
        lib.slam_debug_ba_host_time(S, L.ptr(b.cams), L.ptr(b.Pn, L.i32p), L.ptr(b.Mn, L.i32p), L.ptr(b.On, L.i32p), L.ptr(b.theta0), L.ptr(b.tc, L.u8p), L.ptr(b.px),
                                    L.ptr(b.pi, L.i64p), L.ptr(b.li, L.i64p), t, L.ptr(out))
        best = out.copy() if best is None or out[:2].sum() < best[:2].sum() else best
    print(f"S = {S}, {t} threads: plan {best[0]:.0f} us, emit {best[1]:.0f} us  ({best[0] / S:.1f} + {best[1] / S:.1f} us per window at 1 thread equivalent x{t})")
