"""CPU: known-answer tests for the oracle's bundle adjustment, and a cross-check of
its linearisation against an independent numpy restatement (tests/np_ba.py)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import np_ba


def test_rotzyx_convention(orc):
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.uniform(-1.2, 1.2, 3)
        R = orc.rotzyx(*a)
        assert np.abs(R - Rotation.from_euler("ZYX", a).as_matrix()).max() < 1e-14
        assert np.allclose(orc.rotzyx_angles(R), a, atol=1e-12)
        assert np.abs(R - np_ba.rotzyx(a)).max() < 1e-15


def test_reduced_system_vs_numpy_and_partition_additivity(orc, syn):
    s = syn.ba_scene(P=4, M=60, seed=2, obs_per_point=3)
    P, M, O = s["P"], s["M"], s["O"]
    sh = np_ba.NumpyShard(s["cam"], P, s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    red = sh.build(0, 0.1).numpy()
    n = 6 * P
    S, g, ud, ssr = orc.ba_reduced_system(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"],
                                          np.zeros(O, np.uint8), 0, 0.1, 0, M)
    scale = np.abs(S).max()
    assert np.abs(red[:n * n].reshape(n, n, order="F") - S).max() < 1e-9 * scale
    assert np.abs(red[n * n:n * n + n] - g).max() < 1e-9 * np.abs(g).max()
    assert np.abs(red[n * n + n:n * n + 2 * n] - ud).max() < 1e-9 * np.abs(ud).max()
    assert abs(red[n * n + 2 * n] - ssr) < 1e-9 * ssr
    assert np.abs(S - S.T).max() < 1e-9 * scale
    assert np.array_equal(S[:6, :], np.zeros((6, n)))                          # constant pose: no Jacobian columns
    # shards by point range add up (what the all-reduce relies on)
    parts = [orc.ba_reduced_system(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"],
                                   np.zeros(O, np.uint8), 0, 0.1, a, b) for a, b in ((0, 17), (17, 40), (40, M))]
    assert np.abs(sum(p[0] for p in parts) - S).max() < 1e-9 * scale
    assert abs(sum(p[3] for p in parts) - ssr) < 1e-9 * ssr


@pytest.mark.parametrize("solver", [0, 1])
def test_noise_free_scene_converges_to_ground_truth(orc, syn, solver):
    s = syn.ba_scene(P=5, M=150, seed=3, noise_px=0.0, outlier_frac=0.0)
    th, ol, st = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"],
                                       iters_fast=8, iterations=12, solver=solver)
    assert st["ssr_final"] < 1e-6 * st["ssr_init"]
    assert not ol.any()
    Y = orc.ba_residuals(s["cam"], th, s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    assert abs((Y * Y).sum() - st["ssr_final"]) <= 1e-9 + 1e-9 * st["ssr_final"]
    assert np.array_equal(th[:6], s["theta0"][:6])                             # constant pose untouched


def test_noisy_scene_outliers_and_solver_agreement(orc, syn):
    s = syn.ba_scene(P=6, M=400, seed=4)
    r = {}
    for solver in (0, 1):
        r[solver] = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=solver)
    (t0, o0, s0), (t1, o1, s1) = r[0], r[1]
    assert abs(s0["ssr_final"] - s1["ssr_final"]) < 1e-3 * s1["ssr_final"]
    assert (o0 != o1).mean() < 0.01
    assert np.abs(t0[:36] - t1[:36]).max() < 1e-3
    flagged = set(np.where(o1)[0])
    assert len(flagged & set(s["gross_outliers"])) >= 0.9 * len(s["gross_outliers"])   # 4-12 px off vs the 2.24 px gate
    assert np.abs(t1[:36] - s["theta_gt"][:36]).max() < 0.02
    assert s1["ssr_final"] < s1["ssr_pass1"] < s1["ssr_init"]
    assert s0["inner_iters"] > 0 and s1["inner_iters"] == 0


def test_pnp_ba(orc, syn):
    s = syn.pnp_scene(n=200, seed=0)
    pose, e0, e1, ol, no = orc.pnp_ba(s["cam"], s["pose0"], s["pixels_yx"], s["points"], repr_eps=3.0)
    assert e1 < e0 and np.abs(pose - s["pose_gt"]).max() < 0.02
    assert set(np.where(ol)[0]) >= set(s["gross_outliers"]) and no == ol.sum()
    assert np.abs(pose[:3, :3] @ pose[:3, :3].T - np.eye(3)).max() < 1e-12 and np.array_equal(pose[3], [0, 0, 0, 1])
    px = s["pixels_yx"][:6].copy(); px[:4] += 300
    pose, e0, e1, ol, no = orc.pnp_ba(s["cam"], s["pose0"], px, s["points"][:6], repr_eps=3.0)
    assert 6 - no < 5 and np.array_equal(pose, np.eye(4))                      # bundle_adjustment.jl:157-161
