"""GPU: edge cases of the C ABI (empty / degenerate inputs, error paths)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ba_without_observations_and_single_pose(slam, orc, syn):
    cam = syn.KITTI_CAM
    theta = np.array([0.01, 0.02, 0.03, 0.1, 0.2, 0.3, 1.0, 2.0, 10.0])           # 1 pose, 1 point, no observation
    cache = slam.LocalBACache(theta.copy(), np.array([0], np.uint8), np.zeros((0, 2)), np.zeros(0, np.int64), np.zeros(0, np.int64))
    slam.bundle_adjustment_(cache, cam)
    assert np.array_equal(cache.theta, theta) and len(cache.outliers) == 0
    # one free pose, a handful of points: behaves like PnP + point refinement
    s = syn.ba_scene(P=2, M=30, seed=1, obs_per_point=2, n_const=1)
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(cache, s["cam"])
    th, ol, st = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
    assert np.array_equal(cache.outliers, ol) and np.abs(cache.theta - th).max() < 1e-6


def test_ba_large_window_tile_boundaries(slam, orc, syn):
    """6P = 96 / 102 / 192: reduced systems that end exactly on, just after, and two tiles after a 32-tile boundary."""
    for P in (16, 17, 32):
        s = syn.ba_scene(P=P, M=60 * P, seed=P)
        cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        slam.bundle_adjustment_(cache, s["cam"])
        th, ol, st = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
        assert np.array_equal(cache.outliers, ol), P
        assert np.abs(cache.theta - th).max() <= 1e-6 * max(1.0, np.abs(th).max()), P
        assert abs(cache.stats["ssr_final"] - st["ssr_final"]) <= 1e-8 * st["ssr_final"], P


def _ba_vs_oracle(slam, orc, s, tag):
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(cache, s["cam"])
    th, ol, st = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
    assert np.array_equal(cache.outliers, ol), tag
    assert (cache.stats["iters_pass1"], cache.stats["iters_pass2"]) == (st["iters_pass1"], st["iters_pass2"]), tag
    assert np.abs(cache.theta - th).max() <= 1e-6 * max(1.0, np.abs(th).max()), tag
    assert abs(cache.stats["ssr_final"] - st["ssr_final"]) <= 1e-8 * st["ssr_final"], tag


def test_ba_point_groups_with_shuffled_ids_scattered_constant_poses_and_orphan_points(slam, orc, syn):
    """The grouped build sorts map points by their first FREE observer: shuffle point ids and the observation order, make poses in the
    middle of the window constant (points whose first observer is constant, points seen by constant poses only), add points nobody
    observes -- results must still equal the oracle's."""
    s = syn.ba_scene(P=12, M=700, seed=5, obs_per_point=6, n_const=1)
    rng = np.random.default_rng(3)
    P, M = 12, 700
    const = s["theta_const"].copy(); const[[4, 5, 9]] = 1                       # constant poses inside the window
    perm_pts = rng.permutation(M)                                                # new id of point j = perm_pts[j] + 1
    theta = s["theta0"].copy()
    pts = theta[6 * P:].reshape(M, 3).copy()
    new_pts = np.zeros((M + 3, 3)); new_pts[perm_pts] = pts
    new_pts[M:] = [[1.0, 2.0, 30.0], [-3.0, 0.5, 12.0], [0.0, 0.0, 50.0]]        # three map points without observations
    order = rng.permutation(len(s["pose_ids"]))
    keep = np.ones(len(order), bool)
    # points 0..19 keep only their observations by constant poses (if they have any)
    pid0 = s["point_ids"] - 1
    only_const = (pid0 < 20) & (const[s["pose_ids"] - 1] == 0)
    has_const = np.zeros(M, bool); np.logical_or.at(has_const, pid0, const[s["pose_ids"] - 1] == 1)
    keep &= ~(only_const & has_const[pid0])
    order = order[keep[order]]
    s2 = dict(s)
    s2["theta0"] = np.concatenate([theta[:6 * P], new_pts.ravel()])
    s2["theta_const"] = const
    s2["pose_ids"] = s["pose_ids"][order]; s2["point_ids"] = (perm_pts[pid0[order]] + 1).astype(np.int64)
    s2["pixels_yx"] = s["pixels_yx"][order]
    _ba_vs_oracle(slam, orc, s2, "shuffled")


def test_ba_wide_windows_fall_back_to_pair_lists_and_the_tiled_solve(slam, orc, syn):
    """Every point seen by 24 consecutive key-frames: block half-bandwidth 23 > 20, so the point groups / banded solve do not apply and
    the pair-list build + tiled Cholesky + per-point back-substitution run instead."""
    s = syn.ba_scene(P=26, M=500, seed=11, obs_per_point=24, n_const=1)
    _ba_vs_oracle(slam, orc, s, "wide")
    # a single map point with more observations than a group holds (> 448, almost all from constant poses) also takes that path
    s = syn.ba_scene(P=6, M=200, seed=12, obs_per_point=4, n_const=1)
    rng = np.random.default_rng(1)
    n_extra = 240                                                                 # old key-frames, each observing the point twice
    j = 7                                                                         # the point that every "old key-frame" also sees
    P = 6
    # extra constant poses observing point j (twice each: only free poses must not repeat), appended as poses P+1 .. P+240
    th = s["theta0"]; M = 200
    base_pose = th[:6].copy()
    extra = np.tile(base_pose, (n_extra, 1)) + rng.normal(0, [1e-3] * 3 + [5e-2] * 3, (n_extra, 6))
    Xj = th[6 * P + 3 * j: 6 * P + 3 * j + 3]
    import numpy as _np
    def proj(pose, X):
        t1, t2, t3 = pose[:3]
        Rz = _np.array([[_np.cos(t1), -_np.sin(t1), 0], [_np.sin(t1), _np.cos(t1), 0], [0, 0, 1]])
        Ry = _np.array([[_np.cos(t2), 0, _np.sin(t2)], [0, 1, 0], [-_np.sin(t2), 0, _np.cos(t2)]])
        Rx = _np.array([[1, 0, 0], [0, _np.cos(t3), -_np.sin(t3)], [0, _np.sin(t3), _np.cos(t3)]])
        Xc = Rz @ Ry @ Rx @ X + pose[3:]
        fx, fy, cx, cy = s["cam"]
        return _np.array([fy * Xc[1] / Xc[2] + cy, fx * Xc[0] / Xc[2] + cx])
    px = _np.stack([proj(e, Xj) for e in extra] * 2) + rng.normal(0, 0.5, (2 * n_extra, 2))
    s3 = dict(s)
    s3["theta0"] = np.concatenate([th[:6 * P], extra.ravel(), th[6 * P:]])
    s3["theta_const"] = np.concatenate([s["theta_const"], np.ones(n_extra, np.uint8)])
    s3["pose_ids"] = np.concatenate([s["pose_ids"], np.tile(np.arange(P + 1, P + n_extra + 1), 2)]).astype(np.int64)
    s3["point_ids"] = np.concatenate([s["point_ids"], np.full(2 * n_extra, j + 1)]).astype(np.int64)
    s3["pixels_yx"] = np.concatenate([s["pixels_yx"], px])
    _ba_vs_oracle(slam, orc, s3, "many observers")


def test_ba_loop_closure_windows_are_solved_in_a_folded_pose_order(slam, orc, syn):
    """A window whose first and last key-frames share map points is a ring, not a band: slam_local_ba relabels the poses (fold /
    Cuthill-McKee, constant poses first: tests/test_ba_plan_order.py), solves on the banded path and hands theta back in the caller's
    order.  Against the oracle, which solves in the caller's order: same outliers, iterations, cost, parameters."""
    for P, k_loop, n_const in ((30, 5, 1), (40, 3, 2), (24, 5, 1), (50, 5, 1)):
        s = syn.ba_scene_loop(P=P, M=30 * P, seed=400 + P, n_loop=120, k_loop=k_loop, n_const=n_const)
        cache = slam.LocalBACache(s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        _, hb, reordered = slam.ba_plan_order(cache)
        assert reordered and hb <= 20, (P, hb)
        _ba_vs_oracle(slam, orc, s, ("loop", P))
    # constant poses between the free ones: span 21 in the caller's order, 7 among the free poses
    P = 45
    s = syn.ba_scene(P=P, M=900, seed=44, obs_per_point=22)
    const = np.ones(P, dtype=np.uint8); const[::3] = 0
    s["theta_const"] = const
    cache = slam.LocalBACache(s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    assert slam.ba_plan_order(cache)[1:] == (7, True)
    _ba_vs_oracle(slam, orc, s, "interleaved constants")


def test_ba_ragged_windows_fuzz(slam, orc, syn):
    """Sixty random windows (syn.ba_scene_ragged: dropped observations, constant poses anywhere, loop closures, shuffled order) against
    the oracle -- every solver path the dispatch can take (twisted / single-workgroup / wide band, relabelled poses, pair lists).
    tests/fuzz/ba_fuzz.py runs more of them."""
    for seed in range(60):
        s = syn.ba_scene_ragged(seed)
        _ba_vs_oracle(slam, orc, s, ("ragged", seed))


def test_ba_refuses_bad_ids_and_repeated_observers(slam, syn):
    """an id out of range, or a map point observed twice by one FREE pose (no place in a pose block), is an argument error on both set-up
    paths (the arrays written in place into the pinned block: banded windows; host vectors: the general path); theta stays untouched"""
    import pytest
    for opp in (6, 24):                                                           # banded / general path
        s = syn.ba_scene(P=26, M=300, seed=21, obs_per_point=opp)
        for what in ("pose id", "point id", "twice"):
            pose_ids, point_ids = s["pose_ids"].copy(), s["point_ids"].copy()
            if what == "pose id": pose_ids[17] = 27
            elif what == "point id": point_ids[40] = 0
            else:
                i = int(np.where(s["theta_const"][pose_ids - 1] == 0)[0][5]); j = i + 1 if point_ids[i + 1] == point_ids[i] else i - 1
                pose_ids[j] = pose_ids[i]                                         # same point, same free pose twice
            cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], pose_ids, point_ids)
            with pytest.raises(slam.SlamHipError, match="out of range" if what != "twice" else "observed twice"):
                slam.bundle_adjustment_(cache, s["cam"])
            assert np.array_equal(cache.theta, s["theta0"])
    # twice by a CONSTANT pose is allowed (tests above: "many observers")


def test_ba_twisted_factorisation_split_sizes(slam, orc, syn):
    """Windows of >= max(2 (hb + 1) - 1, hb + 8) free poses are factored from both ends (two workgroups, hb middle poses merged): the smallest such windows
    for three band widths (10, 6 and 2 observers per point), odd / even splits, windows on both sides of the threshold."""
    for P, opp in ((30, 10), (31, 10), (33, 10), (25, 10), (26, 10), (27, 10), (19, 10), (20, 10), (21, 10), (17, 6), (18, 6), (19, 6), (23, 6),
                   (12, 6), (13, 6), (14, 6), (8, 2), (9, 2), (10, 2), (11, 2)):
        s = syn.ba_scene(P=P, M=40 * P, seed=100 + P, obs_per_point=opp)
        _ba_vs_oracle(slam, orc, s, (P, opp))


def test_ba_two_workgroup_solve_is_deterministic(slam, syn):
    """The two sides of the twisted factorisation hand data over through global memory (release / acquire flags carrying the launch
    epoch): the same window solved repeatedly must give bit-identical parameters, outliers and cost (scripts/probes/ba_repeat.py runs more)."""
    s = syn.ba_scene(P=40, M=3000, seed=3)
    ref = None
    for _ in range(40):
        cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        slam.bundle_adjustment_(cache, s["cam"])
        key = (cache.theta.tobytes(), cache.outliers.tobytes(), cache.stats["ssr_final"])
        ref = ref or key
        assert key == ref


def test_ba_failed_factorisation_leaves_the_cache_untouched(slam, syn):
    """A reduced camera system that is not positive definite (here: a NaN map point poisons every block it touches) ends the call with
    SLAM_ERR_NUMERIC and theta / outliers as they were (the reference's LSMR step cannot fail and never leaves cache.theta half
    updated) -- on the single-workgroup solve, on the two-workgroup one (whose sides wait for each other: no hang), and on the tiled
    path of wide windows."""
    for P, opp in ((12, 10), (40, 10), (26, 24)):
        s = syn.ba_scene(P=P, M=60 * P, seed=P, obs_per_point=opp)
        th = s["theta0"].copy(); th[6 * P + 3 * (30 * P) + 1] = np.nan                 # y of a map point in the middle of the window
        cache = slam.LocalBACache(th, s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        before = cache.theta.copy()
        with pytest.raises(slam.SlamHipError) as ei:
            slam.bundle_adjustment_(cache, s["cam"])
        assert "not positive definite" in str(ei.value), (P, str(ei.value))
        assert np.array_equal(cache.theta, before, equal_nan=True), P
        # the context is still usable: the same window without the NaN solves
        good = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
        slam.bundle_adjustment_(good, s["cam"])
        assert good.stats["ssr_final"] < good.stats["ssr_init"], P


def test_error_paths(slam, texture):
    ctx = slam.default_context(0)
    with pytest.raises(slam.SlamHipError):
        slam.LKPyramid(shape=(16, 16), levels=3)                    # level 3 would be 2x2: too small for the 3-pole IIR
    img = texture(70, 105)[0][0]
    e = slam.Extractor(60, 17, (2, 3), 35)
    out = np.zeros((2, 2), dtype=np.int64); n = C.c_int(0)
    from slam_jl_amd import _lib as L
    fimg = np.asfortranarray(img)
    rc = ctx.lib.slam_detect(ctx.h, L.ptr(fimg), 70, 105, None, 0, 60, 17, 2, 3, 35, 3.0, 1e-4, L.ptr(out, L.i64p), 2, C.byref(n))
    assert rc == -4 and b"cap" in ctx.lib.slam_last_error(ctx.h)    # SLAM_ERR_CAPACITY
    bits, rc2 = slam.describe(e, img, np.zeros((0, 2), dtype=np.int64))
    assert len(bits) == 0 and len(rc2) == 0
    bits, rc2 = slam.describe(e, img, np.array([[1, 1], [70, 105]]))            # every keypoint on the border: all dropped
    assert len(bits) == 0
    a = slam.LKPyramid(img, 2); b = slam.LKPyramid(shape=(64, 64), levels=2)
    with pytest.raises(slam.SlamHipError):
        slam.copy_(a, b)                                              # shape mismatch
    new, st = slam.optical_flow_matching(a, a, np.zeros((0, 2)), np.zeros(0, bool), np.zeros((0, 2)), slam.Params(pyramid_levels=2))
    assert len(new) == 0 and len(st) == 0


def test_identical_frames_track_to_zero_motion(slam, texture, orc):
    img = texture(120, 160)[0][0]
    a = slam.LKPyramid(shape=img.shape, levels=3); slam.update_(a, img)
    b = slam.deepcopy(a)
    kp = orc.detect(img, np.zeros((0, 2)), max_points=100).astype(float)
    out, st = slam.fb_tracking_(a, b, kp, window_size=9, pyramid_levels=3, max_distance=1.0)
    ref = orc.pyr_build(img, 3, 1.0, 1)
    ro, rs = orc.fb_tracking(ref, ref, kp, sum_order=1)
    assert np.array_equal(st, rs) and st.mean() > 0.8               # the eigenvalue gate may reject a few at coarse levels
    assert np.abs(out[st] - kp[st]).max() < 1e-9


def test_ba_wide_bands_take_the_banded_solver_too(slam, orc, syn):
    """Half-bandwidths 10 .. 20 stay on the one-launch banded solve, on its other paths: a panel of more than 64 rows (second trip of the
    factor wave), two window slots per set-up thread (hb >= 14), six prefetch entries per lane, the three-phase back-substitution
    with L_kk^-1 read back from the factor store, no twisted split.  11, 15 and 20 observers per point (hb = 10, 14, 19), and the last
    width the grouped build accepts."""
    for P, opp in ((24, 11), (30, 15), (26, 20), (40, 21)):
        s = syn.ba_scene(P=P, M=50 * P, seed=200 + opp, obs_per_point=opp)
        _ba_vs_oracle(slam, orc, s, (P, opp))


def test_ba_every_band_width_of_the_grouped_build(slam, orc, syn):
    """Every half-bandwidth 9 .. 20 (the grouped build folds 8 / 4 / 2 point subsets; at hb = 15 .. 18 a subset is 192 lanes and a
    quarter of the workgroup belongs to no subset -- those widths failed until round 3), and a single workgroup on more than 85 poses
    (damping entries beyond the workgroup's 512 threads)."""
    for opp in range(10, 22):
        s = syn.ba_scene(P=24, M=600, seed=300 + opp, obs_per_point=opp)
        _ba_vs_oracle(slam, orc, s, (24, opp))
    s = syn.ba_scene(P=96, M=2400, seed=333, obs_per_point=12)
    _ba_vs_oracle(slam, orc, s, (96, 12))
