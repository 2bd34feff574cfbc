"""CPU: the oracle's five-point RANSAC (orc_5pt.c) pinned against numpy.roots, the defining constraints of an
essential matrix and ground-truth two-view scenes.  The reference has no test for this path
(RecoverPose.five_point_ransac is un-vendored): parity unpinned, see the file header."""
import numpy as np
from scipy.spatial.transform import Rotation


def _skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0.0]])


def test_poly_real_roots_match_numpy(orc):
    rng = np.random.default_rng(1)
    for k in range(600):
        deg = int(rng.integers(1, 11))
        p = np.poly(rng.normal(size=deg) * 3)[::-1] * rng.normal() if k % 2 else rng.normal(size=deg + 1)
        r = orc.poly_real_roots(p)
        ref = np.roots(p[::-1])
        ref = np.sort(ref[np.abs(ref.imag) < 1e-7 * np.maximum(1, np.abs(ref))].real)
        assert len(r) == len(ref), (p, r, ref)
        assert np.all(np.diff(r) >= 0)
        if len(r):
            assert np.max(np.abs(r - ref) / np.maximum(1, np.abs(ref))) < 1e-6, (p, r, ref)


def test_poly_real_roots_special_cases(orc):
    assert len(orc.poly_real_roots([1, 0, 1.0])) == 0                                # x^2 + 1
    assert np.allclose(orc.poly_real_roots([-2, 1.0]), [2.0])
    assert np.allclose(orc.poly_real_roots([0, 0, 1, 0, 0.0]), [0.0])                # trailing zero coefficients; double root
    assert np.allclose(orc.poly_real_roots(np.poly([1, 2, 3, 4, 5, 6, 7, 8, 9, 10.0])[::-1]), np.arange(1, 11), atol=1e-6)


def test_essential_poses_contain_the_true_motion(orc):
    rng = np.random.default_rng(2)
    for _ in range(50):
        R = Rotation.from_rotvec(rng.normal(size=3) * 0.5).as_matrix()
        t = rng.normal(size=3); t /= np.linalg.norm(t)
        E = _skew(t) @ R * rng.normal() * 10 ** rng.uniform(-2, 2)
        poses = orc.essential_poses(E)
        assert len(poses) == 4
        for P in poses:
            assert np.allclose(P[:, :3] @ P[:, :3].T, np.eye(3), atol=1e-9) and abs(np.linalg.det(P[:, :3]) - 1) < 1e-9
            assert abs(np.linalg.norm(P[:, 3]) - 1) < 1e-12
            En = _skew(P[:, 3]) @ P[:, :3]                                           # each reproduces E up to scale
            assert min(np.abs(En / np.linalg.norm(En) - s * E / np.linalg.norm(E)).max() for s in (1, -1)) < 1e-9
        assert min(np.abs(P - np.c_[R, t]).max() for P in poses) < 1e-9
    assert orc.essential_poses(np.zeros((3, 3))) == []


def test_minimal_solver_satisfies_constraints_and_contains_truth(orc, syn):
    sc = syn.five_point_scene(n=60, noise_px=0.0, outlier_frac=0.0)
    Egt = _skew(sc["Rt_gt"][:, 3]) @ sc["Rt_gt"][:, :3]; Egt /= np.linalg.norm(Egt)
    rng = np.random.default_rng(3)
    found = 0
    for _ in range(300):
        ids = rng.permutation(60)[:5]
        q1, q2 = sc["pd1"][ids], sc["pd2"][ids]
        Es = orc.five_point_solve(q1, q2)
        assert 1 <= len(Es) <= 10
        h1 = np.c_[q1, np.ones(5)]; h2 = np.c_[q2, np.ones(5)]
        for E in Es:
            En = E / np.linalg.norm(E)
            assert np.abs(np.einsum("ij,jk,ik->i", h2, En, h1)).max() < 1e-9           # the five epipolar constraints
        d = [min(np.abs(E / np.linalg.norm(E) - s * Egt).max() for s in (1, -1)) for E in Es]
        k = int(np.argmin(d))
        found += d[k] < 1e-6
        if d[k] < 1e-10:                                                             # a true essential matrix
            En = Es[k] / np.linalg.norm(Es[k])
            assert abs(np.linalg.det(En)) < 1e-9
            assert np.abs(2 * En @ En.T @ En - np.trace(En @ En.T) * En).max() < 1e-8
    assert found >= 0.95 * 300                                                       # the rest: ill-conditioned 5-tuples


def test_minimal_solver_degenerate(orc):
    q = np.array([[0.1, 0.2], [0.1, 0.2], [0.3, -0.1], [0.0, 0.0], [-0.2, 0.1]])     # zero motion, repeated point
    Es = orc.five_point_solve(q, q)
    for E in Es:
        assert np.isfinite(E).all()


def test_ransac_recovers_motion_and_rejects_gross(orc, syn):
    sc = syn.five_point_scene(n=300, seed=2, noise_px=0.3, outlier_frac=0.25, iters=64)
    cnt, E, P, inl, err, bi = orc.five_point_ransac(sc["px1"], sc["px2"], sc["pd1"], sc["pd2"], sc["K"], sc["K"], 3.0, sc["samples"])
    assert cnt == inl.sum() and cnt >= 0.7 * 300
    assert inl[sc["gross"]].mean() < 0.2            # a displacement along the epipolar line is not observable
    assert np.abs(P[:, :3] - sc["Rt_gt"][:, :3]).max() < 2e-2        # a minimal-sample estimate under 0.3 px noise
    assert np.degrees(np.arccos(np.clip(P[:, 3] @ sc["Rt_gt"][:, 3], -1, 1))) < 10.0
    assert abs(np.linalg.norm(P[:, 3]) - 1) < 1e-12 and err > 0 and 0 <= bi < 64
    En = _skew(P[:, 3]) @ P[:, :3]
    assert min(np.abs(En / np.linalg.norm(En) - s * E / np.linalg.norm(E)).max() for s in (1, -1)) < 1e-9


def test_ransac_invalid_samples(orc, syn):
    sc = syn.five_point_scene(n=50, seed=4)
    bad = np.array([[0, 0, 1, 2, 3], [5, 60, 2, 1, 0], [-1, 2, 3, 4, 5]], dtype=np.int32)
    cnt, E, P, inl, err, bi = orc.five_point_ransac(sc["px1"], sc["px2"], sc["pd1"], sc["pd2"], sc["K"], sc["K"], 3.0, bad)
    assert cnt == 0 and bi == -1 and not inl.any() and err == 0.0 and not P.any()


def test_draw_samples_five(slam_host):
    for n in (5, 6, 40, 1000):
        s = slam_host.draw_samples(n, 300, seed=n, k=5)
        assert s.shape == (300, 5) and s.dtype == np.int32 and s.min() >= 0 and s.max() < n
        assert all(len(set(r)) == 5 for r in s.tolist())
    assert slam_host.draw_samples(4, 10, k=5).shape == (0, 5)
