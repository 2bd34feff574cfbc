"""CPU: the oracle's triangulation (orc_tri.c) pinned against numpy (eigh / svd of the DLT matrix) and ground truth.
The reference has no test for this path (RecoverPose is un-vendored): parity unpinned, see the file header."""
import numpy as np
import pytest


def _dlt(P1, P2, a, b):
    x1, y1, x2, y2 = a[1], a[0], b[1], b[0]
    return np.stack([x1 * P1[2] - P1[0], y1 * P1[2] - P1[1], x2 * P2[2] - P2[0], y2 * P2[2] - P2[1]])


def test_min_eigvec_matches_numpy(orc):
    rng = np.random.default_rng(1)
    for _ in range(50):
        A = rng.normal(size=(4, 4)) * rng.uniform(0.1, 1000, size=(4, 1))
        S = A.T @ A
        v = orc.sym4_min_eigvec(S)
        w, V = np.linalg.eigh(S)
        ref = V[:, 0]
        v = v / np.linalg.norm(v) * np.sign(v @ ref)
        gap = (w[1] - w[0]) / w[3]
        assert np.allclose(v, ref, atol=1e-13 / max(gap, 1e-6)), (v, ref, w)


@pytest.mark.parametrize("temporal", [False, True])
def test_triangulation_recovers_ground_truth(orc, syn, slam_host, temporal):
    s = syn.triangulation_scene(n=400, seed=3, temporal=temporal)
    P1, P2 = slam_host.projection_matrices(s["cam"], s["cam"], s["T21"])
    xyz, st = orc.triangulate(P1, P2, s["T21"], s["cam"], s["cam"], s["px1"], s["px2"], max_error=3.0)
    assert st.all()
    assert np.max(np.abs(xyz - s["xyz"]) / np.abs(s["xyz"]).max(axis=1, keepdims=True)) < 1e-7
    # the same homogeneous point as the smallest right singular vector of the DLT matrix
    for i in range(0, 400, 37):
        A = _dlt(P1, P2, s["px1"][i], s["px2"][i])
        v = np.linalg.svd(A)[2][-1]
        assert np.allclose(v[:3] / v[3], xyz[i], rtol=1e-6, atol=1e-9)


def test_gates(orc, syn, slam_host):
    s = syn.triangulation_scene(n=300, seed=5, noise_px=0.3, n_behind=20, n_gross=25)
    P1, P2 = slam_host.projection_matrices(s["cam"], s["cam"], s["T21"])
    xyz, st = orc.triangulate(P1, P2, s["T21"], s["cam"], s["cam"], s["px1"], s["px2"], max_error=3.0)
    assert not st[s["behind"]].any()            # depth gate (left_point[3] < 0.1)
    assert not st[s["gross"]].any()             # reprojection gate
    good = np.setdiff1d(np.arange(300), np.concatenate([s["behind"], s["gross"]]))
    assert st[good].mean() > 0.97
    # temporal semantics: a violated gate only removes the observation when parallax > 20 (mapper.jl:243-258)
    par = np.full(300, 5.0); par[s["gross"][:10]] = 40.0
    _, st_t = orc.triangulate(P1, P2, s["T21"], s["cam"], s["cam"], s["px1"], s["px2"], max_error=3.0, parallax=par, min_parallax=20.0)
    assert not st_t[s["gross"][:10]].any()
    assert st_t[s["gross"][10:]].all() and st_t[s["behind"]].all()


def test_inverse_iteration_eigvec_matches_jacobi_and_numpy(orc, syn, slam_host):
    """orc_sym4_min_eigvec_invit (hypothesis scoring) against the Jacobi routine and numpy on real DLT matrices."""
    s = syn.triangulation_scene(n=600, seed=8, noise_px=0.5, temporal=True)
    P1, P2 = slam_host.projection_matrices(s["cam"], s["cam"], s["T21"])
    worst = 0.0
    for i in range(600):
        A = _dlt(P1, P2, s["px1"][i], s["px2"][i])
        S = A.T @ A
        v1 = orc.sym4_min_eigvec(S); v2 = orc.sym4_min_eigvec(S, inverse_iteration=True)
        X1, X2 = v1[:3] / v1[3], v2[:3] / v2[3]
        worst = max(worst, np.abs(X1 - X2).max() / np.abs(X1).max())
        if i % 50 == 0:
            w, V = np.linalg.eigh(S)
            ref = V[:3, 0] / V[3, 0]
            assert np.allclose(X2, ref, rtol=1e-6)
    assert worst < 1e-7
    v = orc.sym4_min_eigvec(np.zeros((4, 4)), inverse_iteration=True)               # degenerate input: finite output
    assert np.isfinite(v).all()
