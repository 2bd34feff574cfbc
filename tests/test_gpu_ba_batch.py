"""GPU: slam_local_ba_batch -- bundle_adjustment! (src/bundle_adjustment.jl:1-111) for S windows in one set of launches, the estimator
tasks of S lock-stepped SlamManagers (src/estimator.jl:78-99, :317-347).  Every window must come out as S separate slam_local_ba calls
leave it (outlier sets array_equal, cost 1e-8, theta 1e-6 -- in fact bit-equal: same kernels, same order) and as the oracle's Schur-LM
does, on ragged window sizes, with windows that converge early, windows the batch kernels do not cover and degenerate ones mixed in."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cache(slam, s):
    return slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])


def _scenes(syn):
    sc = [syn.ba_scene(P=25, M=800, seed=5, n_const=20),            # the reference's shape: 5 free + 20 constant key-frames
          syn.ba_scene(P=5, M=300, seed=0), syn.ba_scene(P=8, M=600, seed=1), syn.ba_scene(P=20, M=2000, seed=2),
          syn.ba_scene(P=25, M=500, seed=11, n_const=21), syn.ba_scene(P=12, M=150, seed=12, n_const=3),
          syn.ba_scene(P=6, M=40, seed=13), syn.ba_scene(P=30, M=1200, seed=14, n_const=24)]
    return sc


def test_batch_equals_single_calls_and_oracle_on_ragged_windows(slam, orc, syn):
    sc = _scenes(syn)
    single = []
    for s in sc:
        c = _cache(slam, s); slam.bundle_adjustment_(c, s["cam"]); single.append(c)
    caches = [_cache(slam, s) for s in sc]
    status = slam.bundle_adjustment_batch_(caches, [s["cam"] for s in sc])
    assert not status.any(), status
    for z, (s, c, ref) in enumerate(zip(sc, caches, single)):
        assert np.array_equal(c.outliers, ref.outliers), z
        assert c.stats["iters_pass1"] == ref.stats["iters_pass1"] and c.stats["iters_pass2"] == ref.stats["iters_pass2"], z
        for k in ("ssr_init", "ssr_pass1", "ssr_final"):
            assert abs(c.stats[k] - ref.stats[k]) <= 1e-8 * ref.stats[k], (z, k)
        assert np.abs(c.theta - ref.theta).max() <= 1e-6 * max(1.0, np.abs(ref.theta).max()), z
        th, ol, st = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], 5, 10, 5.0, solver=1)
        assert np.array_equal(c.outliers, ol), z
        assert abs(c.stats["ssr_final"] - st["ssr_final"]) <= 1e-8 * st["ssr_final"], z
        assert np.abs(c.theta - th).max() <= 1e-6 * max(1.0, np.abs(th).max()), z
        cst = s["theta_const"].astype(bool); P = len(cst)
        assert np.array_equal(c.theta[:6 * P].reshape(P, 6)[cst], s["theta0"][:6 * P].reshape(P, 6)[cst]), z      # constant poses never move


def test_window_kernel_on_one_and_on_two_workgroups_agree(slam, orc, syn):
    """k_ba_window runs a window on TWO workgroups while both halves of every window fit the chip (<= 128 such windows per call), on one
    otherwise: the same reference-shaped windows in a call of 9 and in a call of 137 (ragged: 5 / 3 / 1 free poses, few points, one
    point) give the same outlier sets and iteration counts, costs and parameters to rounding; first and last against the oracle"""
    base = [syn.ba_scene(P=25, M=800, seed=5, n_const=20), syn.ba_scene(P=25, M=300, seed=6, n_const=22), syn.ba_scene(P=12, M=90, seed=7, n_const=11),
            syn.ba_scene(P=9, M=33, seed=8, n_const=4), syn.ba_scene(P=25, M=500, seed=9, n_const=21), syn.ba_scene(P=6, M=2, seed=10, n_const=3),
            syn.ba_scene(P=7, M=1, seed=11, n_const=2), syn.ba_scene(P=30, M=1000, seed=12, n_const=25), syn.ba_scene(P=10, M=700, seed=13, n_const=5)]
    few = [_cache(slam, s) for s in base]
    st = slam.bundle_adjustment_batch_(few, [s["cam"] for s in base])
    assert not st.any(), st
    many_sc = [base[z % len(base)] for z in range(137)]
    many = [_cache(slam, s) for s in many_sc]
    st = slam.bundle_adjustment_batch_(many, [s["cam"] for s in many_sc])
    assert not st.any(), st
    for z, c in enumerate(many):
        ref = few[z % len(base)]
        assert np.array_equal(c.outliers, ref.outliers), z
        assert c.stats["iters_pass1"] == ref.stats["iters_pass1"] and c.stats["iters_pass2"] == ref.stats["iters_pass2"], z
        assert abs(c.stats["ssr_final"] - ref.stats["ssr_final"]) <= 1e-9 * ref.stats["ssr_final"], z
        assert np.abs(c.theta - ref.theta).max() <= 1e-9 * max(1.0, np.abs(ref.theta).max()), z
    for z in (0, len(base) - 1):
        s = base[z]
        th, ol, so = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], 5, 10, 5.0, solver=1)
        assert np.array_equal(few[z].outliers, ol) and abs(few[z].stats["ssr_final"] - so["ssr_final"]) <= 1e-8 * so["ssr_final"], z
        assert np.abs(few[z].theta - th).max() <= 1e-6 * max(1.0, np.abs(th).max()), z


@pytest.mark.parametrize("ns", [1, 2, 7, 8, 9, 17, 64, 128])
def test_window_kernel_at_every_batch_size(slam, syn, ns):
    """the two-workgroup grid of k_ba_window is 16 x ceil(ns / 8) workgroups (pairs b, b + 8): every window count up to 128 leaves the
    surplus workgroups idle and solves each window as a call of its own does"""
    base = [syn.ba_scene(P=25, M=200 + 40 * z, seed=60 + z, n_const=20 + (z % 3)) for z in range(5)]
    ref = []
    for s in base:
        c = _cache(slam, s); slam.bundle_adjustment_(c, s["cam"]); ref.append(c)
    sc = [base[z % len(base)] for z in range(ns)]
    caches = [_cache(slam, s) for s in sc]
    st = slam.bundle_adjustment_batch_(caches, [s["cam"] for s in sc])
    assert not st.any(), st
    for z, c in enumerate(caches):
        r = ref[z % len(base)]
        assert np.array_equal(c.outliers, r.outliers), z
        assert c.stats["iters_pass1"] == r.stats["iters_pass1"] and c.stats["iters_pass2"] == r.stats["iters_pass2"], z
        assert abs(c.stats["ssr_final"] - r.stats["ssr_final"]) <= 1e-8 * r.stats["ssr_final"], z
        assert np.abs(c.theta - r.theta).max() <= 1e-6 * max(1.0, np.abs(r.theta).max()), z


def test_two_full_batches_at_once_share_the_chip(slam, syn):
    """two contexts launch k_ba_window for 128 windows each at the same time: 2 x 256 workgroups whose halves wait for each other compete
    for 256 compute units -- workgroups are dispatched in order, so a waiting half's partner is at most eight places behind it and every
    pair gets its turn (no deadlock); results as a call alone gives them"""
    import threading
    base = [syn.ba_scene(P=25, M=300, seed=300 + z, n_const=20) for z in range(4)]
    sc = [base[z % 4] for z in range(128)]
    ref = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"]); ref.solve()
    out = {}
    def run(tag):
        ctx = slam.Context(0)
        for _ in range(6):
            b = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"]); b.solve(ctx=ctx)
        out[tag] = b; ctx.close()
    th = [threading.Thread(target=run, args=(t,)) for t in range(3)]
    [t.start() for t in th]; [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th), "a batch did not return within two minutes"
    for t in range(3):
        assert not out[t].status.any()
        assert np.array_equal(out[t].theta, ref.theta) and np.array_equal(out[t].outl, ref.outl), t


def test_window_kernel_on_a_context_confined_to_few_compute_units(slam, syn):
    """a stream confined to 32 compute units (slam_ctx_create_cumask) can hold 32 of k_ba_window's workgroups at once: 16 windows still run on
    two workgroups each (both halves resident), 24 fall back to one -- the halves of a window must never wait for a workgroup that cannot start"""
    base = [syn.ba_scene(P=25, M=250, seed=400 + z, n_const=20) for z in range(3)]
    ctx = slam.Context(0, cu_mask=[1 if c < 32 else 0 for c in range(256)])
    for ns in (16, 24):
        sc = [base[z % 3] for z in range(ns)]
        ref = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"]); ref.solve()
        b = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"]); b.solve(ctx=ctx)
        assert not b.status.any()
        assert np.array_equal(b.outl, ref.outl)
        assert np.abs(b.theta - ref.theta).max() <= 1e-9 * max(1.0, np.abs(ref.theta).max()), ns
    ctx.close()


def test_batch_with_windows_outside_the_batch_kernels(slam, orc, syn):
    """a dense window (half-bandwidth 23: the general path), a loop-closure window (solved on relabelled poses), an all-constant window,
    an empty one and a regular one in the same call"""
    sc = [syn.ba_scene(P=26, M=600, seed=9, obs_per_point=24), syn.ba_scene_loop(P=30, M=1500, seed=7, n_loop=200),
          syn.ba_scene(P=4, M=100, seed=4, n_const=4), syn.ba_scene(P=5, M=200, seed=3)]
    empty = dict(theta0=np.zeros(12), theta_const=np.array([1, 0], dtype=np.uint8), pixels_yx=np.zeros((0, 2)), pose_ids=np.zeros(0, np.int64),
                 point_ids=np.zeros(0, np.int64), cam=sc[0]["cam"])
    sc.insert(2, empty)
    single = []
    for s in sc:
        c = _cache(slam, s); slam.bundle_adjustment_(c, s["cam"]); single.append(c)
    caches = [_cache(slam, s) for s in sc]
    status = slam.bundle_adjustment_batch_(caches, [s["cam"] for s in sc])
    assert not status.any(), status
    for z, (c, ref) in enumerate(zip(caches, single)):
        assert np.array_equal(c.outliers, ref.outliers), z
        assert np.abs(c.theta - ref.theta).max() <= 1e-6 * max(1.0, np.abs(ref.theta).max()), z
        if ref.stats["ssr_final"] > 0:
            assert abs(c.stats["ssr_final"] - ref.stats["ssr_final"]) <= 1e-8 * ref.stats["ssr_final"], z


def test_batch_of_128_reference_shaped_windows(slam, orc, syn):
    """the bench's batch: 128 windows of 5 free + 20 constant key-frames (estimator.jl:327-331), different data per window; windows 0, 64
    and 127 against the oracle, all against each other's iteration counts being plausible"""
    S = 128
    sc = [syn.ba_scene(P=25, M=800, seed=100 + z, n_const=20) for z in range(S)]
    caches = [_cache(slam, s) for s in sc]
    status = slam.bundle_adjustment_batch_(caches, sc[0]["cam"])
    assert not status.any()
    for z in (0, 64, 127):
        s, c = sc[z], caches[z]
        th, ol, st = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], 5, 10, 5.0, solver=1)
        assert np.array_equal(c.outliers, ol), z
        assert abs(c.stats["ssr_final"] - st["ssr_final"]) <= 1e-8 * st["ssr_final"], z
        assert np.abs(c.theta - th).max() <= 1e-6 * max(1.0, np.abs(th).max()), z
    ref = _cache(slam, sc[77]); slam.bundle_adjustment_(ref, sc[77]["cam"])
    # (256-thread workgroups sum a group's points in four subsets instead of eight: equal to rounding, not to the bit)
    assert np.abs(caches[77].theta - ref.theta).max() <= 1e-9 and np.array_equal(caches[77].outliers, ref.outliers)


def test_batch_reports_a_bad_window_and_solves_the_others(slam, syn):
    good = syn.ba_scene(P=5, M=200, seed=3)
    bad = syn.ba_scene(P=5, M=200, seed=4)
    bad = dict(bad); bad["pose_ids"] = bad["pose_ids"].copy(); bad["pose_ids"][7] = 99            # pose id out of range
    caches = [_cache(slam, good), _cache(slam, bad), _cache(slam, good)]
    status = slam.bundle_adjustment_batch_(caches, good["cam"])
    assert status[0] == 0 and status[2] == 0 and status[1] != 0
    ref = _cache(slam, good); slam.bundle_adjustment_(ref, good["cam"])
    assert np.abs(caches[0].theta - ref.theta).max() <= 1e-9 and np.array_equal(caches[2].theta, caches[0].theta)
    assert np.array_equal(caches[1].theta, bad["theta0"])                                          # untouched


def test_batch_calls_from_two_threads_at_once(slam, syn):
    """two estimator threads, each with its own context, call slam_local_ba_batch concurrently (the host half shares one worker pool):
    both get what a serial call gives"""
    import threading
    sc = [syn.ba_scene(P=25, M=400, seed=200 + z, n_const=20) for z in range(6)] + [syn.ba_scene(P=10, M=300, seed=210)]
    ref = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"]); ref.solve()
    out = {}
    def run(tag):
        ctx = slam.Context(0)
        for _ in range(5):
            b = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"]); b.solve(ctx=ctx)
        out[tag] = b; ctx.close()
    th = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    for t in range(2):
        assert not out[t].status.any()
        assert np.array_equal(out[t].theta, ref.theta) and np.array_equal(out[t].outl, ref.outl), t


_XWAIT = r'''
import sys, ctypes, numpy as np
sys.path.insert(0, %(root)r)
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn, _lib
sc = [syn.ba_scene(P=25, M=300, seed=700 + z, n_const=20) for z in range(%(S)d)]
def cache(s):
    return slam.LocalBACache(s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
b = slam.BABatch([cache(s) for s in sc], sc[0]["cam"]); b.solve()
lib = _lib.load(); lib.slam_debug_ba_xretries.restype = ctypes.c_long
print("RETRIES", lib.slam_debug_ba_xretries())
assert not b.status.any(), b.status
np.save(%(out)r, np.concatenate([b.theta.ravel(), b.outl.ravel().astype(np.float64)]))
print("OK")
'''


def test_halves_that_miss_each_other_fall_back_to_one_workgroup(tmp_path):
    """k_ba_window's wait for the partner workgroup is bounded: with the bound at 0 us (SLAMHIP_BA_XWAIT_US=0) a half whose partner has not
    posted yet gives up at once, the window comes back flagged and slam_local_ba_batch solves the call again on one workgroup per window --
    the results are those of SLAMHIP_BA_WINDOW_ONE=1 to the bit and no call hangs"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, env in (("giveup", {"SLAMHIP_BA_XWAIT_US": "0"}), ("one", {"SLAMHIP_BA_WINDOW_ONE": "1"})):
        out = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", _XWAIT % dict(root=root, S=24, out=out)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-800:] + r.stderr[-1500:]
        outs[tag] = (np.load(out), int(r.stdout.split("RETRIES")[1].split()[0]))
    assert outs["giveup"][1] >= 1, "the bound of 0 us never triggered: the test does not exercise the fallback"
    assert outs["one"][1] == 0
    assert np.array_equal(outs["giveup"][0], outs["one"][0])


def test_begin_end_equals_the_blocking_call_and_overlaps_other_work(slam, syn):
    """slam_local_ba_batch_begin / _end: the job runs on the library's thread while the caller does other device work on another context (here:
    a second batch, blocking); the results are the blocking call's to the bit; a second _begin on a busy context and an _end without a job are
    argument errors"""
    sc = [syn.ba_scene(P=20, M=600, seed=900 + z) for z in range(6)] + [syn.ba_scene(P=25, M=300, seed=910, n_const=20)]
    ref = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"]); ref.solve()
    a = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"]); b = slam.BABatch([_cache(slam, s) for s in sc], sc[0]["cam"])
    ctx_a, ctx_b = slam.Context(0), slam.Context(0)
    a.begin(ctx=ctx_a)
    with pytest.raises(slam.SlamHipError, match="already has a batch in flight"):
        b.begin(ctx=ctx_a)
    b.solve(ctx=ctx_b)                                   # other work while the job runs
    st = a.end()
    assert not st.any() and not b.status.any()
    assert np.array_equal(a.theta, ref.theta) and np.array_equal(a.outl, ref.outl) and np.array_equal(b.theta, ref.theta)
    with pytest.raises(slam.SlamHipError, match="no batch in flight"):
        ctx_a.check(ctx_a.lib.slam_local_ba_batch_end(ctx_a.h))
    a.begin(ctx=ctx_a, reset=True)                       # a context that is closed with a job in flight waits for it
    ctx_a.close(); ctx_b.close()
    assert np.array_equal(a.theta, ref.theta)


_MFMA = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
def cache(s):
    return slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
# windows of 8 .. 40 poses whose points are seen by 3 .. 20 key-frames (window half-bandwidths 2 .. 19: 2 .. 8 MFMA tiles per dimension), constant
# poses at the start and ragged ones (dropped observations, constant poses anywhere, loop closures, shuffled order)
sc = [syn.ba_scene(P=12, M=500, seed=1, obs_per_point=3), syn.ba_scene(P=16, M=700, seed=2, obs_per_point=6, n_const=4),
      syn.ba_scene(P=24, M=900, seed=3, obs_per_point=12), syn.ba_scene(P=40, M=600, seed=4, obs_per_point=20, n_const=2),
      syn.ba_scene(P=20, M=4000, seed=5), syn.ba_scene(P=9, M=60, seed=6, obs_per_point=8)]
sc += [syn.ba_scene_ragged(seed=40 + k) for k in range(6)]          # (three of these have a matrix Y beyond 64 KB -- a wide band with few observations per point -- and keep
                                                                     #  the vector kernel, window by window: both builds run in the same launch sets)
b = slam.BABatch([cache(s) for s in sc], sc[0]["cam"]); b.solve()
np.save(%(out)r, np.concatenate([b.theta.ravel(), b.outl.ravel().astype(np.float64), b.stats[:, :6].ravel(), b.status.astype(np.float64)]))
print("OK")
'''


def test_matrix_core_schur_build_equals_the_vector_kernel(tmp_path):
    """k_schur_groups_m (the Schur products as Float64 MFMA tiles, Y = W chol(V^-1)) against the vector kernel k_schur_groups_b (SLAMHIP_BA_NO_MFMA=1) on
    windows of half-bandwidth 2 .. 19 incl. constant poses, ragged observation sets and the bench's P20 shape: outlier sets and iteration counts equal,
    theta to 1e-9 relative, costs to 1e-10 -- only the association of the block products differs"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("mfma", {}), ("vector", {"SLAMHIP_BA_NO_MFMA": "1"})):
        out = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", _MFMA % dict(root=root, out=out)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-800:] + r.stderr[-1500:]
        res[tag] = np.load(out)
    a, b = res["mfma"], res["vector"]
    assert a.shape == b.shape and not np.array_equal(a, b), "the two builds gave bit-identical results: is the matrix-core path taken at all?"
    rel = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert rel.max() <= 1e-9, rel.max()
