"""Random ragged windows through the two-shard emulation of the device-paced sharded LM pass (tests/test_gpu_sharded_emulation.py)
against slam_local_ba on the whole window: python tests/fuzz/ba_shard_fuzz.py [n] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn, sharded_ba
import test_gpu_sharded_emulation as emu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60; s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(s0, s0 + n):
    s = syn.ba_scene_ragged(seed)
    if s["M"] < 4: continue
    rng = np.random.default_rng(seed)
    cut = int(rng.integers(1, s["M"] - 1)) if rng.random() < 0.5 else None
    bounds = [(0, cut), (cut, s["M"])] if cut else sharded_ba.partition_points(s["point_ids"], s["M"], 2)
    tag = f"seed {seed} P {s['P']} M {s['M']} O {s['O']} bounds {bounds}"
    try:
        ref = emu._single(slam, s)
    except slam.SlamHipError as ex:
        print("skip (single solve fails):", tag, str(ex)[:80]); continue
    try:
        theta, outl, st = emu._two_shard_ba(slam, s, bounds)
    except Exception as ex:
        bad += 1
        if isinstance(ex, AssertionError) and len(ex.args) and isinstance(ex.args[0], tuple) and len(ex.args[0]) == 4:
            _, it, a, b = ex.args[0]
            print("FAIL", tag, "shards disagree after iteration", it, "at", np.where(a != b)[0], a[a != b], b[a != b], flush=True)
        else: print("FAIL", tag, repr(ex)[:300], flush=True)
        continue
    msgs = []
    if (st["iters_pass1"], st["iters_pass2"]) != (ref.stats["iters_pass1"], ref.stats["iters_pass2"]): msgs.append(f"iterations {st['iters_pass1']},{st['iters_pass2']} vs {ref.stats['iters_pass1']},{ref.stats['iters_pass2']}")
    if not np.array_equal(outl, ref.outliers): msgs.append(f"outliers differ at {int((outl != ref.outliers).sum())}")
    rel = abs(st["ssr_final"] - ref.stats["ssr_final"]) / ref.stats["ssr_final"]
    if rel > 1e-8: msgs.append(f"ssr rel {rel:.2e}")
    dth = np.abs(theta - ref.theta).max() / max(1.0, np.abs(ref.theta).max())
    if dth > 1e-6: msgs.append(f"theta {dth:.2e}")
    if msgs: bad += 1; print("FAIL", tag, "hbs", st["hbs"], "; ".join(msgs), flush=True)
print(f"{n} windows, {bad} failures")
