"""Random ragged BA windows (observers dropped at random, constant poses anywhere, loop-closure points, shuffled observation order) against
the oracle: python tests/fuzz/ba_fuzz.py [n] [seed0] [batch]
batch > 1: the windows go through slam_local_ba_batch in batches of that many (ragged sizes, reordered and general-path windows mixed in)."""
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


def check_batch(slam, orc, scenes, tags):
    caches = [slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"]) for s in scenes]
    status = slam.bundle_adjustment_batch_(caches, [s["cam"] for s in scenes])
    out = []
    for s, c, tag, stc in zip(scenes, caches, tags, status):
        th, ol, st = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
        if stc != 0:
            out.append(f"{tag}: status {stc} (oracle chol_fail {st.get('chol_fail')})"); continue
        bad = []
        if not np.array_equal(c.outliers, ol): bad.append(f"outliers differ at {int((c.outliers != ol).sum())}")
        if (c.stats["iters_pass1"], c.stats["iters_pass2"]) != (st["iters_pass1"], st["iters_pass2"]): bad.append(f"iterations {c.stats['iters_pass1']},{c.stats['iters_pass2']} vs {st['iters_pass1']},{st['iters_pass2']}")
        dth = np.abs(c.theta - th).max() / max(1.0, np.abs(th).max())
        if dth > 1e-6: bad.append(f"theta {dth:.2e}")
        rel = abs(c.stats["ssr_final"] - st["ssr_final"]) / st["ssr_final"]
        if rel > 1e-8: bad.append(f"ssr rel {rel:.2e}")
        if bad: out.append(f"{tag}: " + "; ".join(bad))
    return out


if __name__ == "__main__":
    import torch
    import slam_jl_amd as slam
    from slam_jl_amd import synthetic as syn
    from oracle import oracle as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100; s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nbad = 0; hbs = []; nre = 0
    nbatch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if nbatch > 1:
        for b0 in range(s0, s0 + n, nbatch):
            scenes = [syn.ba_scene_ragged(seed) for seed in range(b0, min(b0 + nbatch, s0 + n))]
            tags = [f"seed {b0 + k} P {s['P']} free {int((s['theta_const'] == 0).sum())} M {s['M']} O {s['O']}" for k, s in enumerate(scenes)]
            for r in check_batch(slam, orc, scenes, tags):
                nbad += 1; print("FAIL", r, flush=True)
        print(f"{n} windows in batches of {nbatch}, {nbad} failures")
        sys.exit(1 if nbad else 0)
    for seed in range(s0, s0 + n):
        s = syn.ba_scene_ragged(seed)
        order, hb, reordered = slam.ba_plan_order(slam.LocalBACache(s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"]))
        tag = f"seed {seed} P {s['P']} free {int((s['theta_const'] == 0).sum())} M {s['M']} O {s['O']} hb {hb}{' reordered' if reordered else ''}"
        hbs.append(hb); nre += bool(reordered)
        r = check(slam, orc, s, tag)
        if r: nbad += 1; print("FAIL", r, flush=True)
    hbs = np.asarray(hbs)
    print(f"{n} windows, {nbad} failures; reordered {nre}; half-bandwidth <= 9: {int((hbs <= 9).sum())}, 10-20: {int(((hbs > 9) & (hbs <= 20)).sum())}, > 20 (general path): {int((hbs > 20).sum())}")
