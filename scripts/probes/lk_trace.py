"""Phase breakdown of the batched LK kernel (needs slam.jl_amd/libslamhip_trace.so built with -DLK_TRACE)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slam_jl_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libslamhip_trace.so")
sys.argv = [sys.argv[0]] + sys.argv[1:]
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prof_flow.py")).read().split("import ctypes as C, time")[0])
lib = ctx.lib
lib.slam_debug_lk_ticks.restype = C.c_int
buf = (C.c_ulonglong * 8)()
lib.slam_debug_lk_ticks(buf)
n = len(kp)
for its in (0, 1, 30):
    new, ok = slam.optical_flow_matching_batch(pb[0], pb[1], sid, kp, is3d, proj, params, iterations=its, ctx=ctx)
    lib.slam_debug_lk_ticks(buf)
    t = [b * 10.0 / n / 1e3 for b in buf]      # us per point (100 MHz ticks)
    print(f"iterations={its}: per point us: total {t[0]:.2f} | position + offsets {t[5]:.2f} | set-up loads (template, patch, corners) + gradient {t[1]:.2f} | eig test {t[2]:.2f} | LDS samples + accumulate {t[3]:.2f} | reduce + solve {t[4]:.2f}")
