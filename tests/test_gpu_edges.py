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
