"""CPU: known-answer tests for the oracle's Lucas-Kanade / forward-backward tracking."""
import numpy as np
import pytest


def test_svd_pinv_2x2_vs_numpy(orc):
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = rng.normal(size=(2, 2)); G = A @ A.T
        U, S, V = orc.svd2x2(G)
        assert np.abs(U @ np.diag(S) @ V.T - G).max() < 1e-12
        assert np.allclose(np.sort(S)[::-1], np.linalg.svd(G, compute_uv=False), rtol=1e-12)
        Gi, S2 = orc.pinv2x2(G)
        assert np.abs(Gi - np.linalg.pinv(G)).max() < 1e-9 * max(1, np.abs(Gi).max())
    Gi, S = orc.pinv2x2(np.array([[4.0, 2.0], [2.0, 1.0]]))                  # rank 1
    assert np.abs(Gi - np.linalg.pinv(np.array([[4.0, 2.0], [2.0, 1.0]]))).max() < 1e-12
    Gi, S = orc.pinv2x2(np.zeros((2, 2)))
    assert np.array_equal(Gi, np.zeros((2, 2)))


def test_known_translation_and_fb_consistency(orc, texture):
    H, W = 120, 160
    L, R, flows = texture(H, W, n=3, step=(1.3, -2.1))
    p0, p1, p2 = (orc.pyr_build(im, 3, 1.0, 1) for im in L)
    kp = orc.detect(L[0], np.zeros((0, 2)), max_points=200).astype(float)
    out, st = orc.fb_tracking(p0, p1, kp)
    assert st.mean() > 0.7
    d = (out - kp)[st]
    assert np.abs(d.mean(0) - np.array(flows[1])).max() < 0.02 and d.std(0).max() < 0.05
    # two frames ahead = twice the flow, still inside the pyramid's capture range
    out2, st2 = orc.fb_tracking(p0, p2, kp)
    assert np.abs((out2 - kp)[st2].mean(0) - np.array(flows[2])).max() < 0.05
    # the forward-backward check kills tracks into unrelated content
    noise = orc.pyr_build(np.random.default_rng(0).random((H, W)), 3, 1.0, 1)
    _, st3 = orc.fb_tracking(p0, noise, kp)
    assert st3.mean() < 0.3 and st3.mean() < 0.4 * st.mean()
    # textureless source: min eigenvalue test
    flat = orc.pyr_build(np.full((H, W), 0.4), 3, 1.0, 1)
    _, st4 = orc.fb_tracking(flat, flat, kp)
    assert not st4.any()


def test_summation_orders_agree_and_prior(orc, texture):
    L, R, flows = texture(120, 160)
    p0, p1 = orc.pyr_build(L[0], 3, 1.0, 1), orc.pyr_build(L[1], 3, 1.0, 1)
    kp = orc.detect(L[0], np.zeros((0, 2)), max_points=150).astype(float) + 0.37
    o0, s0 = orc.fb_tracking(p0, p1, kp, sum_order=0)
    o1, s1 = orc.fb_tracking(p0, p1, kp, sum_order=1)
    assert (s0 != s1).sum() <= 1 and np.abs(o0[s0 & s1] - o1[s0 & s1]).max() < 1e-9
    oh, sh = orc.fb_tracking(p0, p1, kp, sum_order=2)                        # the parked half-wave experiment's order (scripts/ubench/lk_halfwave.hip.txt): matches no shipped kernel
    assert (s0 != sh).sum() <= 1 and np.abs(o0[s0 & sh] - oh[s0 & sh]).max() < 1e-9
    prior = np.tile(np.array(flows[1]) / 2, (len(kp), 1))
    o2, s2 = orc.fb_tracking(p0, p1, kp, disp0=prior, pyramid_levels=1)
    both = s0 & s2
    assert both.mean() > 0.6 and np.abs(o2[both] - o0[both]).max() < 0.05
    ot, st = orc.fb_tracking(p0, p1, kp, threads=4)                          # OpenMP split changes nothing
    assert np.array_equal(st, s0) and np.array_equal(ot[st], o0[s0])


def test_borders_and_errors(orc, texture):
    H, W = 120, 160
    L = texture(H, W)[0]
    p0, p1 = orc.pyr_build(L[0], 3, 1.0, 1), orc.pyr_build(L[1], 3, 1.0, 1)
    ys = np.array([1, 1.4, 2, 9, H - 9, H - 1, H]); xs = np.array([1, 2.7, 10, W - 9, W])
    pts = np.stack([np.repeat(ys, len(xs)), np.tile(xs, len(ys))], 1)
    out, st = orc.fb_tracking(p0, p1, pts)
    assert np.isfinite(out[st]).all()
    assert (out[st] >= 1).all() and (out[st, 0] <= H).all() and (out[st, 1] <= W).all()
    with pytest.raises(RuntimeError, match="Not enough layers"):
        orc.fb_tracking(orc.pyr_build(L[0], 1, 1.0, 1), orc.pyr_build(L[1], 1, 1.0, 1), pts, pyramid_levels=3)
    out, st = orc.fb_tracking(p0, p1, np.zeros((0, 2)))
    assert len(out) == 0
