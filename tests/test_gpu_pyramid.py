"""GPU parity: device LKPyramid planes vs the CPU oracle -- bit-exact
(pyramid.jl:40-137, lucas_kanade.jl:102-138)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PLANES = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")


def _cmp(slam, orc, img, levels, mode, sigma=1.0):
    if mode == 0:
        lk = slam.LKPyramid(img, levels, sigma=sigma)
    else:
        lk = slam.LKPyramid(shape=img.shape, levels=levels)
        slam.update_(lk, img, sigma=sigma)
    ref = orc.pyr_build(img, levels, sigma, mode)
    for l in range(levels + 1):
        assert lk.level_shape(l) == (ref.Hs[l], ref.Ws[l])
        for name in PLANES:
            g = lk.plane(name, l); r = ref.plane(name, l)
            assert np.array_equal(g, r), (name, l, float(np.abs(g - r).max()))
    return lk


@pytest.mark.parametrize("H,W,levels", [(37, 53, 2), (64, 64, 3), (101, 75, 3), (370, 1226, 3), (376, 1241, 3)])
@pytest.mark.parametrize("mode", [0, 1])
def test_pyramid_planes_bit_exact(slam, orc, texture, H, W, levels, mode):
    _cmp(slam, orc, texture(H, W)[0][0], levels, mode)


def test_pyramid_other_sigma_and_update_reuse(slam, orc, texture):
    L = texture(101, 75)[0]
    lk = _cmp(slam, orc, L[0], 2, 1, sigma=1.7)
    slam.update_(lk, L[1], sigma=1.7)                          # in-place reuse, like front_end.jl:461
    ref = orc.pyr_build(L[1], 2, 1.7, 1)
    for name in PLANES:
        assert np.array_equal(lk.plane(name, 2), ref.plane(name, 2))


def test_pyramid_copy_and_clone(slam, texture):
    L = texture(64, 64)[0]
    a = slam.LKPyramid(L[0], 3); b = slam.LKPyramid(L[1], 3)
    c = slam.deepcopy(a)
    slam.copy_(a, b)                                           # prev <- cur (pyramid.jl:28)
    for name in PLANES:
        assert np.array_equal(a.plane(name, 1), b.plane(name, 1))
        assert not np.array_equal(c.plane(name, 1), b.plane(name, 1))


def test_pyramid_integral_property_full_size(slam, texture):
    """size-independent property at BASELINE size: integral planes are
    non-decreasing along both axes for the two squared planes (sums of
    non-negative smoothed squares up to IIR ringing ~1e-3 of the mean)."""
    img = texture(376, 1241)[0][0]
    lk = slam.LKPyramid(shape=img.shape, levels=3)
    slam.update_(lk, img)
    for name in ("Iyy", "Ixx"):
        I = lk.plane(name, 0)
        assert I[-1, -1] > 0
        assert np.isfinite(I).all()
    assert np.array_equal(lk.plane("layers", 0), img)


@pytest.mark.parametrize("H,W", [(37, 53), (101, 75), (370, 1226), (1080, 1920)])
def test_fast_mode_within_tolerance(slam, orc, texture, H, W):
    """mode 3 (segmented recurrences): same planes up to the rounding of the segment
    entry states.  Tolerance: 1e-11 relative to the plane's max magnitude."""
    L = texture(H, W)[0]
    levels = 3 if min(H, W) > 60 else 2
    lk = slam.LKPyramid(shape=(H, W), levels=levels)
    slam.update_(lk, L[0], fast=True)
    ref = orc.pyr_build(L[0], levels, 1.0, 1)
    worst = 0.0
    for l in range(levels + 1):
        for name in PLANES:
            g = lk.plane(name, l); r = ref.plane(name, l)
            err = np.abs(g - r).max() / max(np.abs(r).max(), 1e-300)
            worst = max(worst, err)
            assert err <= 1e-11, (name, l, err)
    # and tracking on fast pyramids agrees with tracking on exact pyramids far below the reference's own eps (1e-2 px)
    lk2 = slam.LKPyramid(shape=(H, W), levels=levels); slam.update_(lk2, L[1], fast=True)
    e1 = slam.LKPyramid(shape=(H, W), levels=levels); slam.update_(e1, L[0])
    e2 = slam.LKPyramid(shape=(H, W), levels=levels); slam.update_(e2, L[1])
    kp = orc.detect(L[0], np.zeros((0, 2)), max_points=500).astype(float)
    of, sf = slam.fb_tracking_(lk, lk2, kp, window_size=9, pyramid_levels=levels, max_distance=1.0)
    oe, se = slam.fb_tracking_(e1, e2, kp, window_size=9, pyramid_levels=levels, max_distance=1.0)
    assert (sf != se).sum() <= max(1, len(kp) // 200)
    both = sf & se
    assert np.abs(of[both] - oe[both]).max() < 1e-7


def test_u8_ingest_bit_exact(slam, orc, texture):
    """8-bit frames (what the KITTI reader decodes): raw/255 on the device == Gray{Float64}.(img) on the host."""
    img = texture(101, 75)[0][0]
    u8 = np.asfortranarray(np.round(img * 255).astype(np.uint8))
    lk = slam.LKPyramid(shape=u8.shape, levels=2)
    slam.update_(lk, u8)
    ref = orc.pyr_build(np.asfortranarray(u8.astype(np.float64) / 255.0), 2, 1.0, 1)
    for name in PLANES:
        for l in range(3):
            assert np.array_equal(lk.plane(name, l), ref.plane(name, l)), (name, l)


def test_chain_replay_is_bit_identical(slam, orc, texture):
    """SLAM_PYR_CHAIN (the build replayed as one chain, no forked integral-image branch) changes the schedule, not one bit of a plane;
    three such builds kept in flight on three contexts (what a host does for frames t+1 .. t+3) equal the oracle as well."""
    import torch
    H, W = 370, 1226
    L, R, flows = texture(H, W, n=3)
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in L]
    torch.cuda.synchronize()
    ctxs = [slam.Context(0) for _ in range(3)]
    pyr = [slam.LKPyramid(shape=(H, W), levels=3, ctx=ctxs[k]) for k in range(3)]
    for rep in range(2):                                                     # capture, then replay
        for k in range(3):
            slam.update_(pyr[k], None, device_ptr=dev[k].data_ptr(), sync=False, ctx=ctxs[k], chain=True)
    for c in ctxs:
        c.synchronize()
    for k in range(3):
        ref = orc.pyr_build(L[k], 3, 1.0, 1)
        for l in range(4):
            for name in ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx"):
                assert np.array_equal(pyr[k].plane(name, l), ref.plane(name, l)), (k, name, l)
    for c in ctxs:
        c.close()
