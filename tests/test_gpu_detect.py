"""GPU parity: slam_detect (HIP) vs the CPU oracle -- keypoint indices bit-exact,
same order (extractor.jl:63-95)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _extractor(slam, H, W, max_points=1000, cell=35):
    return slam.Extractor(max_points, max(5, cell // 2), (-(-H // cell), -(-W // cell)), cell)


@pytest.mark.parametrize("H,W,maxp", [(70, 105, 100), (83, 131, 200), (200, 300, 300), (370, 1226, 1000), (376, 1241, 2000)])
def test_detect_no_mask_exact(slam, orc, texture, H, W, maxp):
    img = texture(H, W)[0][0]
    e = _extractor(slam, H, W, maxp)
    got = slam.detect(e, img, np.zeros((0, 2)))
    ref = orc.detect(img, np.zeros((0, 2)), max_points=maxp, radius=e.radius, cell_size=e.cell_size)
    assert len(ref) > 0
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("H,W,maxp,ncur", [(83, 131, 200, 17), (200, 300, 300, 60), (370, 1226, 1000, 400), (376, 1241, 1000, 999)])
def test_detect_with_avoidance_mask_exact(slam, orc, texture, H, W, maxp, ncur):
    img = texture(H, W)[0][0]
    rng = np.random.default_rng(5)
    cur = np.stack([rng.uniform(1, H, ncur), rng.uniform(1, W, ncur)], 1)
    cur[0] = (1.0, 1.0); cur[1] = (H, W); cur[2] = (0.5, W + 0.49)     # corners / rounding to the border
    e = _extractor(slam, H, W, maxp)
    got = slam.detect(e, img, cur)
    ref = orc.detect(img, cur, max_points=maxp, radius=e.radius, cell_size=e.cell_size)
    assert np.array_equal(got, ref)
    # no new keypoint inside an avoidance disk centre
    if len(got):
        d = np.abs(got[:, None, :] - np.rint(cur)[None, :, :]).max(-1)
        assert d.min() > 0


@pytest.mark.parametrize("cell", [12, 17, 23, 40, 51])
def test_detect_other_cell_sizes_exact(slam, orc, texture, cell):
    """The stencil phases work on strips of five rows / columns: cell sizes that are not multiples of five, smaller than a strip run
    and with ragged border cells (H, W not multiples of the cell), with and without the avoidance mask."""
    H, W = 131, 203
    img = texture(H, W)[0][0]
    e = _extractor(slam, H, W, 400, cell)
    rng = np.random.default_rng(cell)
    cur = np.stack([rng.uniform(1, H, 40), rng.uniform(1, W, 40)], 1)
    for c in (np.zeros((0, 2)), cur):
        got = slam.detect(e, img, c)
        ref = orc.detect(img, c, max_points=400, radius=e.radius, cell_size=cell)
        assert len(ref) > 0 and np.array_equal(got, ref), (cell, len(c))


def test_detect_sigma_zero_and_clustered_points(slam, orc, texture):
    H, W = 200, 300
    img = texture(H, W)[0][0]
    rng = np.random.default_rng(7)
    cur = np.stack([rng.uniform(80, 120, 700), rng.uniform(100, 180, 700)], 1)   # > 512 candidates for one cell
    e = _extractor(slam, H, W, 1000)
    for sig in (0.0, 3.0, 1.5):
        got = slam.detect(e, img, cur, sigma_mask=sig)
        ref = orc.detect(img, cur, max_points=1000, radius=e.radius, cell_size=e.cell_size, sigma_mask=sig)
        assert np.array_equal(got, ref), sig


def test_detect_early_out_and_flat_image(slam, orc):
    H, W = 120, 160
    e = _extractor(slam, H, W, 50)
    img = np.full((H, W), 0.5)
    assert len(slam.detect(e, img, np.zeros((0, 2)))) == 0                      # no response anywhere
    cur = np.ones((50, 2)) * 20
    assert len(slam.detect(e, np.random.default_rng(0).random((H, W)), cur)) == 0   # extractor.jl:64


def test_detect_fhd_4000(slam, orc, texture):
    H, W = 1080, 1920
    img = texture(H, W)[0][0]
    e = _extractor(slam, H, W, 4000)
    rng = np.random.default_rng(11)
    cur = np.stack([rng.uniform(1, H, 1500), rng.uniform(1, W, 1500)], 1)
    got = slam.detect(e, img, cur)
    ref = orc.detect(img, cur, max_points=4000, radius=e.radius, cell_size=e.cell_size)
    assert len(ref) > 1000 and np.array_equal(got, ref)


def test_detect_from_resident_pyramid(slam, orc, texture):
    H, W = 200, 300
    img = texture(H, W)[0][0]
    e = _extractor(slam, H, W, 300)
    lk = slam.LKPyramid(img, 3)
    assert np.array_equal(slam.detect(e, lk, np.zeros((0, 2))), slam.detect(e, img, np.zeros((0, 2))))


def test_detect_idempotent_full_size_property(slam, texture):
    """size-independent property at BASELINE size: re-detecting with the found
    keypoints as avoidance points returns nothing within radius of them."""
    H, W = 376, 1241
    img = texture(H, W)[0][0]
    e = _extractor(slam, H, W, 1000)
    k1 = slam.detect(e, img, np.zeros((0, 2)))
    k2 = slam.detect(e, img, k1[: 600].astype(float))
    if len(k2):
        d2 = ((k2[:, None, :] - k1[None, :600, :]) ** 2).sum(-1)
        assert d2.min() >= 17 ** 2 * 0.5
