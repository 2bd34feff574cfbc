"""CPU: pin the oracle's image primitives against independent numpy/scipy
restatements and analytic known answers (the reference ships no golden
vectors).  Parity with Julia itself stays unpinned."""
import numpy as np
import pytest
from scipy import ndimage


def test_gaussian_taps(orc):
    for sigma, n in ((3.0, 13), (np.sqrt(2.0), 9), (1.0, 5)):
        w = orc.gaussian_taps(sigma)
        x = np.arange(n) - n // 2
        g = np.exp(-x * x / (2 * sigma * sigma))
        assert len(w) == n and np.allclose(w, g / g.sum(), rtol=1e-15)
        assert abs(w.sum() - 1) < 1e-15


@pytest.mark.parametrize("border,mode", [(0, "nearest"), (1, "constant")])
def test_separable_correlation_vs_scipy(orc, border, mode):
    rng = np.random.default_rng(0)
    img = rng.random((23, 31))
    k1 = rng.random(5); k2 = rng.random(7)
    got = orc.imfilter_sep(img, k1, k2, border)
    ref = ndimage.correlate1d(ndimage.correlate1d(img, k1, axis=0, mode=mode), k2, axis=1, mode=mode)
    assert np.abs(got - ref).max() < 1e-14


def _iir_reference_line(x, sigma, orc, pad):
    """Same recursion on an explicitly replicate-padded signal, no boundary algebra."""
    a, scale, M, asum = orc.iir_coeffs(sigma)
    xp = np.concatenate([np.full(pad, x[0]), x, np.full(pad, x[-1])])
    w = np.zeros_like(xp); w[:3] = xp[0] / (1 - asum)
    for i in range(3, len(xp)):
        w[i] = xp[i] + a[0] * w[i - 1] + a[1] * w[i - 2] + a[2] * w[i - 3]
    v = np.zeros_like(xp); v[-3:] = xp[-1] / (1 - asum) ** 2
    for i in range(len(xp) - 4, -1, -1):
        v[i] = w[i] + a[0] * v[i + 1] + a[1] * v[i + 2] + a[2] * v[i + 3]
    return (v * scale)[pad:-pad]


@pytest.mark.parametrize("sigma", [1.0, 4.0])
def test_iir_triggs_sdika_boundaries(orc, sigma):
    """The closed-form boundary initialisation must equal filtering the infinitely
    replicate-extended signal (that is what Triggs & Sdika derive)."""
    rng = np.random.default_rng(1)
    x = rng.random(40)
    img = np.tile(x[:, None], (1, 5))                     # filter along dim 1 only matters; dim 2 is constant
    got = orc.iir_gaussian(img, sigma, 0)
    ref = _iir_reference_line(x, sigma, orc, 600)
    assert np.abs(got[:, 2] - ref).max() < 1e-10
    gotT = orc.iir_gaussian(img.T.copy(), sigma, 0)       # same along dim 2
    assert np.abs(gotT[2, :] - ref).max() < 1e-10


def test_iir_is_a_unit_gain_gaussian(orc):
    img = np.full((30, 41), 0.37)
    assert np.abs(orc.iir_gaussian(img, 4.0, 0) - 0.37).max() < 1e-13
    assert np.abs(orc.iir_gaussian(img, 1.0, 2) - 0.37).max() < 1e-13     # NA(): normalised at the borders
    f0 = orc.iir_gaussian(img, 1.0, 1)                                    # Fill(0): decays at the borders
    assert f0[0, 0] < 0.25 and abs(f0[15, 20] - 0.37) < 1e-6
    rng = np.random.default_rng(2)
    tex = ndimage.gaussian_filter(rng.random((80, 90)), 1.0)
    for sigma, tol in ((4.0, 2e-2), (1.0, 2e-2)):   # 3rd-order recursive approximation of a Gaussian, ~1-2 % of range
        d = orc.iir_gaussian(tex, sigma, 0) - ndimage.gaussian_filter(tex, sigma, mode="nearest", truncate=6.0)
        assert np.abs(d[12:-12, 12:-12]).max() < tol


def test_imresize_and_bilinear(orc):
    rng = np.random.default_rng(3)
    img = rng.random((12, 18))
    half = orc.imresize(img, 6, 9)
    box = 0.25 * (img[0::2, 0::2] + img[1::2, 0::2] + img[0::2, 1::2] + img[1::2, 1::2])
    assert np.abs(half - box).max() < 1e-15
    odd = rng.random((13, 21))
    got = orc.imresize(odd, 7, 11)
    yy = (13 / 7) * (np.arange(1, 8) - 0.5) + 0.5 - 1
    xx = (21 / 11) * (np.arange(1, 12) - 0.5) + 0.5 - 1
    ref = ndimage.map_coordinates(odd, np.meshgrid(yy, xx, indexing="ij"), order=1)
    assert np.abs(got - ref).max() < 1e-14
    for r, c in ((1.0, 1.0), (13.0, 21.0), (4.25, 7.75), (12.999, 1.5)):
        ref = ndimage.map_coordinates(odd, [[r - 1], [c - 1]], order=1)[0]
        assert abs(orc.bilinear(odd, r, c) - ref) < 1e-14


def test_get_mask_disk_rule(orc):
    m = orc.get_mask(60, 70, [[30.4, 35.5], [1.0, 1.0], [2.5, 69.5]], 17)      # 35.5 -> 36, 2.5 -> 2, 69.5 -> 70 (half-even)
    yy, xx = np.mgrid[1:61, 1:71]
    ref = np.ones((60, 70))
    for cy, cx in ((30, 36), (1, 1), (2, 70)):
        ref[((yy - cy) / 17.0) ** 2 + ((xx - cx) / 17.0) ** 2 < 1] = 0
    assert np.array_equal(m, ref)
    assert m[30 - 1, 36 - 1 + 16] == 0 and m[30 - 1, 36 - 1 + 17] == 1        # radius itself is outside (strict <)


def _shi_tomasi_np(cell):
    d = np.array([-0.5, 0.0, 0.5]); s = np.array([0.25, 0.5, 0.25]); b = np.full(3, 1 / 3)
    c = lambda a, k, ax: ndimage.correlate1d(a, k, axis=ax, mode="nearest")
    g1 = c(c(cell, d, 0), s, 1); g2 = c(c(cell, s, 0), d, 1)
    xx = c(c(g1 * g1, b, 0), b, 1); xy = c(c(g1 * g2, b, 0), b, 1); yy = c(c(g2 * g2, b, 0), b, 1)
    return ((xx + yy) - np.sqrt((xx - yy) ** 2 + 4 * xy ** 2)) / 2


def test_shi_tomasi_response_vs_numpy(orc):
    rng = np.random.default_rng(4)
    cell = ndimage.gaussian_filter(rng.random((35, 26)), 1.5)
    assert np.abs(orc.shi_tomasi(cell) - _shi_tomasi_np(cell)).max() < 1e-15
    # an ideal corner has a positive response at the corner and ~0 on the edges
    img = np.zeros((35, 35)); img[17:, 17:] = 1.0
    r = orc.shi_tomasi(img)
    assert r[17, 17] > 1e-3 and abs(r[30, 17]) < 1e-12 and abs(r[5, 5]) < 1e-12


def test_detect_order_quota_and_edge_cases(orc):
    rng = np.random.default_rng(5)
    H, W = 83, 131                                          # ragged: 3 x 4 cells, last row/col truncated
    img = ndimage.gaussian_filter(rng.random((H, W)), 1.2)
    kp = orc.detect(img, np.zeros((0, 2)), max_points=60)   # quota ceil(60/12) = 5 per cell
    cell = (kp - 1) // 35
    cid = cell[:, 0] * 4 + cell[:, 1]
    assert (np.diff(cid) >= 0).all()                         # cells row-major (extractor.jl:81)
    assert np.bincount(cid, minlength=12).max() <= 5
    for c in np.unique(cid):
        k = kp[cid == c]
        lin = (k[:, 1] - 1) * H + k[:, 0]
        assert (np.diff(lin) > 0).all()                      # column-major inside a cell
    assert kp[:, 0].min() >= 1 and kp[:, 0].max() <= H and kp[:, 1].max() <= W
    # independent recomputation of one interior cell
    r = _shi_tomasi_np(img[35:70, 35:70])
    pad = np.pad(r, 1, constant_values=-np.inf)
    nb = np.stack([pad[1 + dy:36 + dy, 1 + dx:36 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)])
    ismax = (r > nb).all(0)
    cand = np.argwhere(ismax)
    cand = cand[np.lexsort((cand[:, 0], cand[:, 1]))]        # column-major
    vals = r[cand[:, 0], cand[:, 1]]
    top = cand[np.argsort(-vals, kind="stable")[:5]]
    top = top[r[top[:, 0], top[:, 1]] >= 1e-4]
    top = top[np.lexsort((top[:, 0], top[:, 1]))] + 1 + 35
    assert np.array_equal(kp[cid == 5], top)
    # early-out, flat image, avoidance
    assert len(orc.detect(img, np.ones((60, 2)), max_points=60)) == 0
    assert len(orc.detect(np.full((H, W), 0.3), np.zeros((0, 2)), max_points=60)) == 0
    k2 = orc.detect(img, kp[:10].astype(float), max_points=60)
    d = np.abs(k2[:, None, :] - kp[None, :10, :]).max(-1)
    assert d.min() > 3


def test_describe_brief(orc):
    rng = np.random.default_rng(6)
    img = ndimage.gaussian_filter(rng.random((64, 80)), 1.0)
    pat = np.clip(np.rint(rng.normal(0, 9 / 5, (256, 4))), -4, 4).astype(np.int32)
    kp = np.array([[1, 1], [5, 5], [6, 6], [32, 40], [59, 75], [60, 40]])
    bits, rc = orc.describe(img, kp, pat)
    assert np.array_equal(rc, [[6, 6], [32, 40], [59, 75]])                  # ceil(9/2) = 5 px border dropped
    taps = orc.gaussian_taps(np.sqrt(2.0))
    sm = ndimage.correlate1d(ndimage.correlate1d(img, taps, axis=0, mode="nearest"), taps, axis=1, mode="nearest")
    y, x = 32 - 1, 40 - 1
    ref = np.array([sm[y + p[0], x + p[1]] < sm[y + p[2], x + p[3]] for p in pat])
    got = np.array([(int(bits[1][b // 64]) >> (b % 64)) & 1 for b in range(256)], dtype=bool)
    assert np.array_equal(got, ref)


def test_pyramid_planes_vs_composition(orc):
    rng = np.random.default_rng(7)
    img = ndimage.gaussian_filter(rng.random((37, 53)), 1.0)
    img = np.asfortranarray(img)
    for mode in (0, 1):
        p = orc.pyr_build(img, 2, 1.0, mode)
        assert (p.Hs, p.Ws) == ([37, 19, 10], [53, 27, 14])
        assert np.array_equal(p.plane("layers", 0), img)
        L1 = orc.imresize(orc.iir_gaussian(img, 1.0, 2 if mode == 0 else 0), 19, 27)
        assert np.array_equal(p.plane("layers", 1), L1)
        d = np.array([-0.5, 0, 0.5]); s = np.array([3, 10, 3]) / 16
        Iy = orc.imfilter_sep(L1, d, s, 1 if mode == 0 else 0); Ix = orc.imfilter_sep(L1, s, d, 1 if mode == 0 else 0)
        assert np.array_equal(p.plane("Iy", 1), Iy) and np.array_equal(p.plane("Ix", 1), Ix)
        Iyx = np.cumsum(np.cumsum(orc.iir_gaussian(Iy * Ix, 4.0, 0), axis=0), axis=1)
        assert np.abs(p.plane("Iyx", 1) - Iyx).max() < 1e-12
    # Scharr of a ramp: interior derivative = slope
    ramp = np.asfortranarray(np.tile(0.01 * np.arange(40)[:, None], (1, 30)))
    p = orc.pyr_build(ramp, 0, 1.0, 1)
    assert np.abs(p.plane("Iy", 0)[5:-5, 5:-5] - 0.01).max() < 1e-15 and np.abs(p.plane("Ix", 0)).max() < 1e-15
