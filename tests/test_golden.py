"""Committed regression vectors (tests/golden/hotpath_v1.npz, oracle-generated --
see make_golden.py for why they cannot be reference-generated) against the
oracle on CPU and against the HIP path on the GPU."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_v1.npz"))
IMG0 = np.asfortranarray(G["img0_u8"].astype(np.float64) / 255)
IMG1 = np.asfortranarray(G["img1_u8"].astype(np.float64) / 255)


def test_oracle_reproduces_golden(orc):
    assert np.array_equal(orc.detect(IMG0, np.zeros((0, 2)), max_points=60), G["kp_nomask"])
    assert np.array_equal(orc.detect(IMG0, G["cur"], max_points=60), G["kp_mask"])
    p0 = orc.pyr_build(IMG0, 2, 1.0, 1); p1 = orc.pyr_build(IMG1, 2, 1.0, 1); pc = orc.pyr_build(IMG0, 2, 1.0, 0)
    assert np.array_equal(p0.plane("Iy", 1), G["upd_Iy_l1"]) and np.array_equal(p0.plane("Iyx", 2), G["upd_Iyx_l2"])
    assert np.array_equal(p0.plane("layers", 2), G["upd_layer_l2"])
    assert np.array_equal(pc.plane("Ixx", 1), G["ctor_Ixx_l1"]) and np.array_equal(pc.plane("layers", 1), G["ctor_layer_l1"])
    out, st = orc.fb_tracking(p0, p1, G["kp_nomask"].astype(float), sum_order=1, pyramid_levels=2)
    assert np.array_equal(st, G["lk_status"]) and np.allclose(out[st], G["lk_out"][st], rtol=0, atol=1e-12)
    th, ol, stats = orc.bundle_adjustment(tuple(G["ba_cam"]), G["ba_theta0"], G["ba_const"], G["ba_pixels"], G["ba_pose_ids"], G["ba_point_ids"], solver=1)
    assert np.array_equal(ol, G["ba_outliers"]) and np.allclose(th, G["ba_theta"], rtol=0, atol=1e-10)
    bits, rc = orc.describe(IMG0, G["kp_nomask"], G["brief_pattern"])
    assert np.array_equal(bits, G["brief_bits"]) and np.array_equal(rc, G["brief_rc"])


@pytest.mark.gpu
def test_hip_reproduces_golden(slam):
    H, W = IMG0.shape
    e = slam.Extractor(60, 17, (-(-H // 35), -(-W // 35)), 35)
    assert np.array_equal(slam.detect(e, IMG0, np.zeros((0, 2))), G["kp_nomask"])
    assert np.array_equal(slam.detect(e, IMG0, G["cur"]), G["kp_mask"])
    p0 = slam.LKPyramid(shape=(H, W), levels=2); slam.update_(p0, IMG0)
    p1 = slam.LKPyramid(shape=(H, W), levels=2); slam.update_(p1, IMG1)
    pc = slam.LKPyramid(IMG0, 2)
    assert np.array_equal(p0.plane("Iy", 1), G["upd_Iy_l1"]) and np.array_equal(p0.plane("Iyx", 2), G["upd_Iyx_l2"])
    assert np.array_equal(p0.plane("layers", 2), G["upd_layer_l2"])
    assert np.array_equal(pc.plane("Ixx", 1), G["ctor_Ixx_l1"]) and np.array_equal(pc.plane("layers", 1), G["ctor_layer_l1"])
    out, st = slam.fb_tracking_(p0, p1, G["kp_nomask"].astype(float), window_size=9, pyramid_levels=2, max_distance=1.0)
    assert np.array_equal(st, G["lk_status"]) and np.abs(out[st] - G["lk_out"][st]).max() < 1e-9
    cache = slam.LocalBACache(G["ba_theta0"].copy(), G["ba_const"], G["ba_pixels"], G["ba_pose_ids"], G["ba_point_ids"])
    slam.bundle_adjustment_(cache, tuple(G["ba_cam"]))
    assert np.array_equal(cache.outliers, G["ba_outliers"]) and np.abs(cache.theta - G["ba_theta"]).max() < 1e-6
    assert abs(cache.stats["ssr_final"] - G["ba_ssr"][2]) < 1e-8 * G["ba_ssr"][2]
    bits, rc = slam.describe(e, IMG0, G["kp_nomask"], pattern=G["brief_pattern"])
    assert np.array_equal(bits, G["brief_bits"]) and np.array_equal(rc, G["brief_rc"])


GP = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_v1.npz"))


def test_oracle_reproduces_pose_golden(orc, slam_host):
    P1, P2 = slam_host.projection_matrices(GP["tri_cam"], GP["tri_cam"], GP["tri_T21"])
    xyz, st = orc.triangulate(P1, P2, GP["tri_T21"], GP["tri_cam"], GP["tri_cam"], GP["tri_px1"], GP["tri_px2"], 3.0)
    assert np.array_equal(st, GP["tri_status"]) and np.array_equal(xyz, GP["tri_xyz"])
    cnt, KP, Rt, inl, err, bi = orc.p3p_ransac(GP["p3p_pts"], GP["p3p_px"], GP["p3p_pdn"], GP["p3p_K"], 3.0, GP["p3p_samples"])
    assert cnt == GP["p3p_n"] and bi == GP["p3p_best"] and err == GP["p3p_error"]
    assert np.array_equal(inl, GP["p3p_inliers"]) and np.array_equal(Rt, GP["p3p_Rt"]) and np.array_equal(KP, GP["p3p_KP"])
    cnt, E, P, inl, err, bi = orc.five_point_ransac(GP["fp_px1"], GP["fp_px2"], GP["fp_pd1"], GP["fp_pd2"], GP["fp_K"], GP["fp_K"], 3.0, GP["fp_samples"])
    assert cnt == GP["fp_n"] and bi == GP["fp_best"] and err == GP["fp_error"]
    assert np.array_equal(inl, GP["fp_inliers"]) and np.array_equal(P, GP["fp_P"]) and np.array_equal(E, GP["fp_E"])


@pytest.mark.gpu
def test_hip_reproduces_pose_golden(slam):
    xyz, st = slam.triangulate(GP["tri_cam"], GP["tri_cam"], GP["tri_T21"], GP["tri_px1"], GP["tri_px2"], 3.0)
    assert np.array_equal(st, GP["tri_status"])
    assert np.max(np.abs(xyz - GP["tri_xyz"]) / np.abs(GP["tri_xyz"]).max(axis=1, keepdims=True)) < 1e-12
    cnt, (KP, inl, err, Rt, bi) = slam.p3p_ransac(GP["p3p_pts"], GP["p3p_px"], GP["p3p_pdn"], GP["p3p_K"], threshold=3.0,
                                                    samples=GP["p3p_samples"], return_pose=True)
    assert cnt == GP["p3p_n"] and bi == GP["p3p_best"] and err == GP["p3p_error"]
    assert np.array_equal(inl, GP["p3p_inliers"]) and np.array_equal(Rt, GP["p3p_Rt"]) and np.array_equal(KP, GP["p3p_KP"])
    cnt, (E, P, inl, err, bi) = slam.five_point_ransac(GP["fp_px1"], GP["fp_px2"], GP["fp_pd1"], GP["fp_pd2"], GP["fp_K"], GP["fp_K"],
                                                        max_repr_error=3.0, samples=GP["fp_samples"], return_extra=True)
    assert cnt == GP["fp_n"] and bi == GP["fp_best"] and err == GP["fp_error"]
    assert np.array_equal(inl, GP["fp_inliers"]) and np.array_equal(P, GP["fp_P"]) and np.array_equal(E, GP["fp_E"])


def test_frontend_fixture_is_what_the_oracle_computes(orc):
    """tests/golden/frontend_v1.npz (inputs of the Julia pin route for optical_flow_matching! + triangulate_stereo!) still equals the oracle."""
    from slam_jl_amd.triangulation import projection_matrices
    F = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_v1.npz"))
    f = lambda u8: np.asfortranarray(u8.astype(np.float64) / 255.0)
    H, W = F["l0_u8"].shape
    cam = tuple(F["cam"])
    p0, p1, pr = (orc.pyr_build(f(F[k]), 3, 1.0, 1) for k in ("l0_u8", "l1_u8", "r1_u8"))
    t = orc.optical_flow_matching(p0, p1, F["kp"], F["is3d"], F["proj"], (H, W), sum_order=0)
    assert np.array_equal(t["updated"], F["t_updated"]) and np.array_equal(t["removed"], F["t_removed"]) and np.array_equal(t["new_pixels"], F["t_new"])
    keep = ~t["removed"]
    kp1, is3d1 = t["new_pixels"][keep], F["is3d"][keep]
    s = orc.optical_flow_matching(p1, pr, kp1, is3d1, F["s_proj"], (H, W), stereo=True, undistorted_left=kp1, right_cam=cam, sum_order=0)
    assert np.array_equal(s["updated"], F["s_updated"]) and np.array_equal(s["removed"], F["s_removed"]) and np.array_equal(s["new_pixels"], F["s_new"])
    sk = ~s["removed"]
    T21 = np.eye(4); T21[0, 3] = -float(F["baseline"][0])
    P1, P2 = projection_matrices(cam, cam, T21)
    kp2, up, syx, i3 = kp1[sk], s["updated"][sk], s["new_pixels"][sk], is3d1[sk]
    cand = np.flatnonzero(up & ~i3)
    assert np.array_equal(cand, F["tri_cand"])
    xyz, ok = orc.triangulate(P1, P2, T21, cam, cam, kp2[cand], syx[cand], 3.0)
    assert np.array_equal(ok, F["tri_ok"]) and np.array_equal(xyz, F["tri_xyz"])
