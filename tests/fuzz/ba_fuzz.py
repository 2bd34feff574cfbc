"""Random ragged BA windows (observers dropped at random, constant poses anywhere, loop-closure points, shuffled observation order) against
the oracle: python tests/fuzz/ba_fuzz.py [n] [seed0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def check(slam, orc, s, tag):
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    th, ol, st = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
    try:
        slam.bundle_adjustment_(cache, s["cam"])
    except slam.SlamHipError as ex:
        return f"{tag}: {str(ex)[:120]} (oracle chol_fail {st.get('chol_fail')})"
    bad = []
    if not np.array_equal(cache.outliers, ol): bad.append(f"outliers differ at {int((cache.outliers != ol).sum())}")
    if (cache.stats["iters_pass1"], cache.stats["iters_pass2"]) != (st["iters_pass1"], st["iters_pass2"]): bad.append(f"iterations {cache.stats['iters_pass1']},{cache.stats['iters_pass2']} vs {st['iters_pass1']},{st['iters_pass2']}")
    dth = np.abs(cache.theta - th).max() / max(1.0, np.abs(th).max())
    if dth > 1e-6: bad.append(f"theta {dth:.2e}")
    rel = abs(cache.stats["ssr_final"] - st["ssr_final"]) / st["ssr_final"]
    if rel > 1e-8: bad.append(f"ssr rel {rel:.2e}")
    return f"{tag}: " + "; ".join(bad) if bad else None


if __name__ == "__main__":
    import torch
    import slam_jl_amd as slam
    from slam_jl_amd import synthetic as syn
    from oracle import oracle as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100; s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nbad = 0; hbs = []; nre = 0
    for seed in range(s0, s0 + n):
        s = syn.ba_scene_ragged(seed)
        order, hb, reordered = slam.ba_plan_order(slam.LocalBACache(s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"]))
        tag = f"seed {seed} P {s['P']} free {int((s['theta_const'] == 0).sum())} M {s['M']} O {s['O']} hb {hb}{' reordered' if reordered else ''}"
        hbs.append(hb); nre += bool(reordered)
        r = check(slam, orc, s, tag)
        if r: nbad += 1; print("FAIL", r, flush=True)
    hbs = np.asarray(hbs)
    print(f"{n} windows, {nbad} failures; reordered {nre}; half-bandwidth <= 9: {int((hbs <= 9).sum())}, 10-20: {int(((hbs > 9) & (hbs <= 20)).sum())}, > 20 (general path): {int((hbs > 20).sum())}")
