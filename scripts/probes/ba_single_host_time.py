#!/usr/bin/env python3
"""Host-side set-up of ONE large window (slam_local_ba's way: the passes over the observations split into tasks of the worker pool; no GPU needed):
python scripts/probes/ba_single_host_time.py [P] [M] [tasks ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import _lib as L, synthetic as syn
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100
M = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
tasks = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8, 12, 32]
lib = slam.load()
s = syn.ba_scene(P=P, M=M, seed=8)
b = slam.BABatch([slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])], s["cam"])
lib.slam_debug_ba_host_time.restype = C.c_int
out = np.zeros(3)
for t in tasks:
    best = None
    for _ in range(15):
        lib.slam_debug_ba_host_time(1, L.ptr(b.cams), L.ptr(b.Pn, L.i32p), L.ptr(b.Mn, L.i32p), L.ptr(b.On, L.i32p), L.ptr(b.theta0), L.ptr(b.tc, L.u8p), L.ptr(b.px),
                                    L.ptr(b.pi, L.i64p), L.ptr(b.li, L.i64p), -t, L.ptr(out))
        best = out.copy() if best is None or out[:2].sum() < best[:2].sum() else best
    print(f"P {P} O {int(b.On[0])}, {t} tasks: plan {best[0]:.0f} us, emit {best[1]:.0f} us, hash {int(best[2])}")
