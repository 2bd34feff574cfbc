"""CPU: the oracle's P3P RANSAC (orc_p3p.c) pinned against numpy.roots, the defining equations and ground truth.
The reference has no test for this path (RecoverPose.p3p_ransac is un-vendored): parity unpinned, see the file header."""
import numpy as np


def test_quartic_matches_numpy_roots(orc):
    rng = np.random.default_rng(1)
    for _ in range(1500):
        A = rng.normal(size=5) * 10 ** rng.uniform(-2, 2, 5)
        r = np.sort(orc.quartic_real_roots(A))
        ref = np.roots(A[::-1])
        ref = np.sort(ref[np.abs(ref.imag) < 1e-9 * np.maximum(1, np.abs(ref))].real)
        assert len(r) == len(ref), (A, r, ref)
        if len(r):
            assert np.max(np.abs(r - ref) / np.maximum(1, np.abs(ref))) < 1e-8, (A, r, ref)


def test_quartic_special_cases(orc):
    assert len(orc.quartic_real_roots([1, 0, 0, 0, 1.0])) == 0                       # x^4 + 1
    r = np.sort(orc.quartic_real_roots([4, 0, -5, 0, 1.0]))                          # (x^2-1)(x^2-4), biquadratic
    assert np.allclose(r, [-2, -1, 1, 2], atol=1e-12)
    r = np.sort(orc.quartic_real_roots(np.poly([0.5, 1.5, -3.0, 7.0])[::-1]))
    assert np.allclose(r, [-3, 0.5, 1.5, 7], atol=1e-11)
    assert len(orc.quartic_real_roots([1, 2, 3, 4, 0.0])) == 0                       # not a quartic: rejected


def test_minimal_solver_contains_true_pose(orc, syn):
    sc = syn.p3p_scene(n=60, noise_px=0.0, outlier_frac=0.0)
    rng = np.random.default_rng(3)
    for _ in range(300):
        ids = rng.permutation(60)[:3]
        sols = orc.p3p_solve(sc["pts3d"][ids], sc["pdn"][ids])
        assert 1 <= len(sols) <= 4
        assert min(np.abs(s - sc["Rt_gt"]).max() for s in sols) < 1e-6
        for s in sols:                                                               # every solution is a rigid pose
            R = s[:, :3]                                                             # that maps the 3 points onto
            assert np.allclose(R @ R.T, np.eye(3), atol=1e-9)                        # their rays
            assert abs(np.linalg.det(R) - 1) < 1e-9
            Y = sc["pts3d"][ids] @ R.T + s[:, 3]
            Yn = Y / np.linalg.norm(Y, axis=1, keepdims=True)
            assert np.allclose(Yn, sc["pdn"][ids], atol=1e-7)


def test_minimal_solver_degenerate(orc):
    X = np.array([[0, 0, 5.0], [1, 0, 5], [2, 0, 5]])                                # collinear world points
    F = X / np.linalg.norm(X, axis=1, keepdims=True)
    assert orc.p3p_solve(X, F) == []
    X = np.array([[0, 0, 5.0], [0, 0, 5.0], [1, 1, 6]])                              # repeated point
    F = X / np.linalg.norm(X, axis=1, keepdims=True)
    assert orc.p3p_solve(X, F) == []


def test_ransac_recovers_pose_and_rejects_gross(orc, syn):
    sc = syn.p3p_scene(n=400, seed=2, noise_px=0.3, outlier_frac=0.25)
    cnt, KP, Rt, inl, err, bi = orc.p3p_ransac(sc["pts3d"], sc["px_xy"], sc["pdn"], sc["K"], 3.0, sc["samples"])
    assert cnt == inl.sum() and cnt >= 0.7 * 400
    assert not inl[sc["gross"]].any()
    assert np.abs(Rt[:, :3] - sc["Rt_gt"][:, :3]).max() < 5e-3 and np.abs(Rt[:, 3] - sc["Rt_gt"][:, 3]).max() < 0.1
    assert np.allclose(KP, sc["K"] @ Rt, atol=1e-9)
    # error = summed reprojection error of the inliers
    Y = sc["pts3d"] @ Rt[:, :3].T + Rt[:, 3]
    uv = (Y[:, :2] / Y[:, 2:]) * [sc["K"][0, 0], sc["K"][1, 1]] + [sc["K"][0, 2], sc["K"][1, 2]]
    e = np.linalg.norm(uv - sc["px_xy"], axis=1)
    assert np.array_equal(inl, e < 3.0)
    assert abs(err - e[inl].sum()) < 1e-8
    assert 0 <= bi < len(sc["samples"])


def test_ransac_invalid_samples_and_no_model(orc, syn):
    sc = syn.p3p_scene(n=50, seed=4)
    bad = np.array([[0, 0, 1], [5, 60, 2], [-1, 2, 3]], dtype=np.int32)
    cnt, KP, Rt, inl, err, bi = orc.p3p_ransac(sc["pts3d"], sc["px_xy"], sc["pdn"], sc["K"], 3.0, bad)
    assert cnt == 0 and bi == -1 and not inl.any() and err == 0.0 and not KP.any()


def test_draw_samples_are_distinct(slam_host):
    for n in (3, 4, 17, 1000):
        s = slam_host.draw_samples(n, 500, seed=n)
        assert s.shape == (500, 3) and s.dtype == np.int32
        assert s.min() >= 0 and s.max() < n
        assert (s[:, 0] != s[:, 1]).all() and (s[:, 0] != s[:, 2]).all() and (s[:, 1] != s[:, 2]).all()
    assert slam_host.draw_samples(2, 10).shape == (0, 3)
