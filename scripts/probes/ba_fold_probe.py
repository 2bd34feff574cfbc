"""Does relabelling the poses of a loop-closure window (fold ordering 0, P-1, 1, P-2, ...) bring it into the banded solver's range, and what
does an LM iteration cost then?  python scripts/ba_fold_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
ctx = slam.Context(0)
def run(s, tag):
    best = None
    for _ in range(4):
        cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        slam.bundle_adjustment_(cache, s["cam"], ctx=ctx)
        it = cache.stats["iters_pass1"] + cache.stats["iters_pass2"]
        ms = cache.stats["device_ms"] / max(it, 1)
        best = ms if best is None else min(best, ms)
    print(f"{tag}: hb {syn.ba_halfband(s)}  iters {it}  {best:.4f} ms/iter  ssr_final {cache.stats['ssr_final']:.9e}", flush=True)
    return cache
def relabel(s, order):
    """order[new] = old (0-based)"""
    P = s["P"]; order = np.asarray(order); new_of = np.empty(P, dtype=np.int64); new_of[order] = np.arange(P)
    t = dict(s)
    th = s["theta0"].copy(); th[:6 * P] = s["theta0"][:6 * P].reshape(P, 6)[order].ravel()
    t["theta0"] = th; t["theta_const"] = np.asarray(s["theta_const"])[order].copy()
    t["pose_ids"] = new_of[s["pose_ids"] - 1] + 1
    return t
for P, M in ((50, 10000),):
    s = syn.ba_scene_loop(P=P, M=M, seed=7, n_loop=1500)
    c0 = run(s, "identity")
    fold = [i // 2 if i % 2 == 0 else P - 1 - i // 2 for i in range(P)]
    try:
        run(syn.ba_scene(P=P, M=4000, seed=3, obs_per_point=19), "plain chain, 19 observers")
    except Exception as ex: print("plain 19:", repr(ex)[:200])
    t = relabel(s, fold)
    if os.environ.get("PROBE_SORT"):
        o = np.lexsort((t["pose_ids"], t["point_ids"]))
        for k in ("pose_ids", "point_ids"): t[k] = t[k][o]
        t["pixels_yx"] = np.ascontiguousarray(t["pixels_yx"][o])
    c1 = run(t, "fold")
    a = np.asarray(c0.theta)[:6 * P].reshape(P, 6)[fold]; b = np.asarray(c1.theta)[:6 * P].reshape(P, 6)
    print("max |pose diff| between the two orderings", np.abs(a - b).max())
    run(syn.ba_scene(P=P, M=M, seed=7), "plain P50")
