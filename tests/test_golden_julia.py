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
    if "prim_box3_of_gy2_cell11" in J.files:                       # round-4 generator: the pieces behind shi_tomasi / findlocalmaxima / boxdiff
        c11 = img0[:35, :35]
        sob_d, sob_s = np.array([-1.0, 0.0, 1.0]) / 2, np.array([1.0, 2.0, 1.0]) / 4
        gy = orc.imfilter_sep(c11, sob_d, sob_s)                   # derivative along dim 1, smoothing along dim 2
        assert np.allclose(gy, J["prim_sobel_y_cell11"], rtol=0, atol=1e-15)
        assert np.allclose(orc.imfilter_sep(c11, sob_s, sob_d), J["prim_sobel_x_cell11"], rtol=0, atol=1e-15)
        third = np.full(3, 1.0 / 3)
        assert np.array_equal(orc.imfilter_sep(gy * gy, third, third), J["prim_box3_of_gy2_cell11"]), "the 3 x 3 box mean is not the separable 1/3 (x) 1/3: keypoint indices can differ"
        resp = orc.shi_tomasi(c11)
        pad = np.pad(resp, 1, constant_values=-np.inf)
        nb = np.stack([pad[1 + dy:36 + dy, 1 + dx:36 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)])
        mx = np.argwhere((resp[None] > nb).all(0).T)[:, ::-1] + 1    # column-major order, 1-based (row, col)
        assert np.array_equal(mx, J["prim_localmaxima_cell11"])
        vals = resp[mx[:, 0] - 1, mx[:, 1] - 1]
        assert np.array_equal(np.argsort(-vals, kind="stable") + 1, J["prim_localmaxima_order_cell11"])
        sch_d, sch_s = np.array([-1.0, 0.0, 1.0]) / 2, np.array([3.0, 10.0, 3.0]) / 16
        assert np.allclose(orc.imfilter_sep(img0, sch_d, sch_s, border=1), J["prim_scharr_fill0_y"], rtol=0, atol=1e-15)
        assert np.allclose(orc.imfilter_sep(img0, sch_s, sch_d, border=1), J["prim_scharr_fill0_x"], rtol=0, atol=1e-15)
        ii = np.cumsum(np.cumsum(img0, axis=0), axis=1)
        bd = lambda y1, y2, x1, x2: ii[y2 - 1, x2 - 1] - (ii[y2 - 1, x1 - 2] if x1 > 1 else 0.0) - (ii[y1 - 2, x2 - 1] if y1 > 1 else 0.0) + (ii[y1 - 2, x1 - 2] if (y1 > 1 and x1 > 1) else 0.0)
        want = [bd(1, 5, 1, 7), bd(3, 21, 1, 19), bd(1, 19, 4, 22), bd(10, 28, 15, 33), bd(H - 18, H, W - 18, W)]
        assert np.allclose(want, J["prim_boxdiff"], rtol=1e-13, atol=0)
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


# ---- part 2: the rows SURVEY 8f added (tests/golden/julia_frontend_v1.npz, written by the same Julia script) -------------------------
JF = os.path.join(HERE, "golden", "julia_frontend_v1.npz")
needs_frontend = pytest.mark.skipif(not os.path.exists(JF), reason="tests/golden/julia_frontend_v1.npz not generated (needs Julia + SLAM.jl: tests/golden/make_golden_julia.jl)")


@needs_frontend
def test_oracle_flow_matching_and_triangulation_match_julia(orc):
    """optical_flow_matching! (temporal + stereo) and triangulate_stereo! of the real SLAM.jl on the inputs of frontend_v1.npz."""
    from slam_jl_amd.triangulation import projection_matrices
    F = np.load(os.path.join(HERE, "golden", "frontend_v1.npz")); J = np.load(JF)
    f = lambda u8: np.asfortranarray(u8.astype(np.float64) / 255.0)
    H, W = F["l0_u8"].shape
    cam = tuple(F["cam"])
    p0, p1, pr = (orc.pyr_build(f(F[k]), 3, 1.0, 1) for k in ("l0_u8", "l1_u8", "r1_u8"))
    t = orc.optical_flow_matching(p0, p1, F["kp"], F["is3d"], F["proj"], (H, W), sum_order=0)
    keep = ~t["removed"]
    assert np.array_equal(keep, J["t_present"].astype(bool))
    assert np.abs(t["new_pixels"][keep] - J["t_new"][keep]).max() <= LK_TOL
    kp1, is3d1 = t["new_pixels"][keep], F["is3d"][keep]
    s = orc.optical_flow_matching(p1, pr, kp1, is3d1, F["s_proj"], (H, W), stereo=True, undistorted_left=kp1, right_cam=cam, sum_order=0)
    assert np.array_equal(~s["removed"], J["s_present"].astype(bool))
    up = s["updated"]
    assert np.array_equal(up, J["s_stereo"].astype(bool))
    assert np.abs(s["new_pixels"][up] - J["s_right"][up]).max() <= LK_TOL
    T21 = np.eye(4); T21[0, 3] = -float(F["baseline"][0])
    P1, P2 = projection_matrices(cam, cam, T21)
    sk = ~s["removed"]
    cand = np.flatnonzero(up & ~is3d1 & sk)
    xyz, ok = orc.triangulate(P1, P2, T21, cam, cam, kp1[cand], s["new_pixels"][cand], 3.0)
    assert np.array_equal(ok, J["tri_is3d"].astype(bool)[cand])
    assert np.abs(xyz[ok] - J["tri_xyz"][cand][ok]).max() <= 1e-6 * max(1.0, np.abs(xyz[ok]).max())      # frame.wc = I: world = camera


@needs_frontend
def test_oracle_pose_primitives_match_recoverpose(orc):
    """RecoverPose.triangulate / p3p_ransac / five_point_ransac on the inputs of pose_v1.npz.  The RANSACs draw their own samples in
    Julia, so poses are compared up to the scenes' noise and inlier sets by overlap; triangulate is deterministic."""
    from slam_jl_amd.triangulation import projection_matrices
    G = np.load(os.path.join(HERE, "golden", "pose_v1.npz")); J = np.load(JF)
    cam = tuple(G["tri_cam"])
    P1, P2 = projection_matrices(cam, cam, G["tri_T21"])
    xyz, st = orc.triangulate(P1, P2, G["tri_T21"], cam, cam, G["tri_px1"], G["tri_px2"], 1e9, min_depth=-1e9)     # no gates: the raw DLT points
    h = J["rp_triangulate_h"]
    ref = h[:, :3] / h[:, 3:4]
    assert np.abs(xyz - ref).max() <= 1e-6 * np.abs(ref).max()
    if "p3p_KP" in J.files:
        cnt, KP, Rt, inl, err, bi = orc.p3p_ransac(G["p3p_pts"], G["p3p_px"], G["p3p_pdn"], G["p3p_K"], 3.0, G["p3p_samples"])
        jin = J["p3p_inliers"].astype(bool)
        assert (inl & jin).sum() >= 0.95 * max(inl.sum(), jin.sum())
        a, b = KP / np.linalg.norm(KP[:, :3]), J["p3p_KP"] / np.linalg.norm(J["p3p_KP"][:, :3])
        assert np.abs(a - b).max() <= 2e-2
    cnt, E, P, inl, err, bi = orc.five_point_ransac(G["fp_px1"], G["fp_px2"], G["fp_pd1"], G["fp_pd2"], G["fp_K"], G["fp_K"], 3.0, G["fp_samples"])
    jin = J["fp_inliers"].astype(bool)
    assert (inl & jin).sum() >= 0.9 * max(inl.sum(), jin.sum())
    Rj, tj = J["fp_P"][:, :3], J["fp_P"][:, 3]
    assert np.abs(P[:, :3] - Rj).max() <= 2e-2 and abs(abs(P[:, 3] @ tj) / (np.linalg.norm(P[:, 3]) * np.linalg.norm(tj)) - 1.0) <= 2e-2
