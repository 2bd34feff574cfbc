"""Reference-pinned parity, active as soon as tests/golden/julia_v1.npz exists.

That file is written by tests/golden/make_golden_julia.jl, which a maintainer with Julia runs against the REAL
SLAM.jl (detect / LKPyramid / update! / fb_tracking! / bundle_adjustment! / describe) on the inputs of
hotpath_v1.npz.  Julia is absent from the build container, so the file is not committed yet and these tests skip;
until they run, parity with the reference is argued from source ("parity unpinned")."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
JPATH = os.path.join(HERE, "golden", "julia_v1.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(JPATH), reason="tests/golden/julia_v1.npz not generated (needs Julia + SLAM.jl: tests/golden/make_golden_julia.jl)")

LK_TOL = 1e-6          # px: the bar of tests/test_gpu_lk.py against the reference's summation order
BA_COST_RTOL = 1e-3    # exact Schur-LM step vs the reference's inexact LM+LSMR step


def _load():
    G = np.load(os.path.join(HERE, "golden", "hotpath_v1.npz"))
    J = np.load(JPATH)
    img0 = np.asfortranarray(G["img0_u8"].astype(np.float64) / 255)
    img1 = np.asfortranarray(G["img1_u8"].astype(np.float64) / 255)
    return G, J, img0, img1


def test_oracle_matches_julia(orc):
    G, J, img0, img1 = _load()
    H, W = img0.shape
    # primitives first: a failure here names the upstream semantic that differs
    assert np.array_equal(orc.get_mask(H, W, G["cur"], 17), J["prim_mask"])
    assert np.allclose(orc.shi_tomasi(img0[:35, :35]), J["prim_shi_tomasi_cell11"], rtol=0, atol=1e-15)
    assert np.allclose(orc.iir_gaussian(img0, 1.0, border=0), J["prim_iir_sigma1_replicate"], rtol=0, atol=1e-14)
    assert np.allclose(orc.iir_gaussian(img0, 4.0, border=0), J["prim_iir_sigma4_replicate"], rtol=0, atol=1e-14)
    assert np.allclose(orc.imresize(img0, -(-H // 2), -(-W // 2)), J["prim_imresize_half"], rtol=0, atol=1e-15)
    # seams
    assert np.array_equal(orc.detect(img0, np.zeros((0, 2)), max_points=60), J["kp_nomask"])
    assert np.array_equal(orc.detect(img0, G["cur"], max_points=60), J["kp_mask"])
    p0 = orc.pyr_build(img0, 2, 1.0, 1); p1 = orc.pyr_build(img1, 2, 1.0, 1); pc = orc.pyr_build(img0, 2, 1.0, 0)
    for name in ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx"):
        for l in range(3):
            assert np.allclose(p0.plane(name, l), J[f"upd_full_{name}_l{l}"], rtol=1e-12, atol=1e-13), (name, l)
    assert np.allclose(pc.plane("Ixx", 1), J["ctor_Ixx_l1"], rtol=1e-12, atol=1e-13)
    assert np.allclose(pc.plane("layers", 1), J["ctor_layer_l1"], rtol=1e-12, atol=1e-13)
    out, st = orc.fb_tracking(p0, p1, J["kp_nomask"].astype(float), sum_order=0, pyramid_levels=2)
    jst = J["lk_status"].astype(bool)
    assert np.array_equal(st, jst) and np.abs(out[st] - J["lk_out"][st]).max() <= LK_TOL
    th, ol, stats = orc.bundle_adjustment(tuple(G["ba_cam"]), G["ba_theta0"], G["ba_const"], G["ba_pixels"], G["ba_pose_ids"], G["ba_point_ids"], solver=0)
    assert (ol != J["ba_outliers"].astype(bool)).mean() <= 0.01
    assert abs(stats["ssr_final"] - J["ba_ssr_final"][0]) <= BA_COST_RTOL * J["ba_ssr_final"][0]
    bits, rc = orc.describe(img0, J["kp_nomask"], J["brief_pattern"])
    assert np.array_equal(rc, J["brief_rc"]) and np.array_equal(bits, J["brief_bits"])


@pytest.mark.gpu
def test_hip_matches_julia(slam):
    G, J, img0, img1 = _load()
    H, W = img0.shape
    e = slam.Extractor(60, 17, (-(-H // 35), -(-W // 35)), 35)
    assert np.array_equal(slam.detect(e, img0, np.zeros((0, 2))), J["kp_nomask"])
    assert np.array_equal(slam.detect(e, img0, G["cur"]), J["kp_mask"])
    p0 = slam.LKPyramid(shape=(H, W), levels=2); slam.update_(p0, img0)
    p1 = slam.LKPyramid(shape=(H, W), levels=2); slam.update_(p1, img1)
    for name in ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx"):
        for l in range(3):
            assert np.allclose(p0.plane(name, l), J[f"upd_full_{name}_l{l}"], rtol=1e-12, atol=1e-13), (name, l)
    out, st = slam.fb_tracking_(p0, p1, J["kp_nomask"].astype(float), window_size=9, pyramid_levels=2, max_distance=1.0)
    jst = J["lk_status"].astype(bool)
    assert np.array_equal(st, jst) and np.abs(out[st] - J["lk_out"][st]).max() <= LK_TOL
    cache = slam.LocalBACache(G["ba_theta0"].copy(), G["ba_const"], G["ba_pixels"], G["ba_pose_ids"], G["ba_point_ids"])
    slam.bundle_adjustment_(cache, tuple(G["ba_cam"]))
    assert (cache.outliers != J["ba_outliers"].astype(bool)).mean() <= 0.01
    assert abs(cache.stats["ssr_final"] - J["ba_ssr_final"][0]) <= BA_COST_RTOL * J["ba_ssr_final"][0]
    bits, rc = slam.describe(e, img0, J["kp_nomask"], pattern=J["brief_pattern"])
    assert np.array_equal(rc, J["brief_rc"]) and np.array_equal(bits, J["brief_bits"])
