"""GPU: the tolerance-mode BATCH build (slam_pyr_update_batch*_dev mode 3, S >= 4: k_cols_fused<TOL> + k_rows_tol -- one read and one
write per plane in the dim-2 stage) against the CPU oracle's exact build (pyramid.jl:81-137, lucas_kanade.jl:109-138).

north_star allows a stated float tolerance for everything except keypoint indices (which come from detect on the raw image).  Bars:
every plane within 1e-11 of the oracle relative to the plane's largest magnitude; tracked positions within 1e-6 px and at most 0.5 %
status flips against the oracle's sequential-order fb_tracking on its exact planes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PLANES = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")
TOL = 1e-11


def _frames(syn, H, W, S, seed):
    rng = np.random.default_rng(seed)
    base = syn.texture_canvas(H, W, seed=seed, margin=0)
    return [np.asfortranarray(np.round(np.clip(base + 0.03 * rng.standard_normal((H, W)), 0, 1) * 255).astype(np.uint8)) for _ in range(S)]


def _check_planes(pyr, ref, levels, tag):
    worst = 0.0
    for l in range(levels + 1):
        for name in PLANES:
            g, r = pyr.plane(name, l), ref.plane(name, l)
            err = float(np.abs(g - r).max() / max(np.abs(r).max(), 1e-300))
            assert err <= TOL, (tag, name, l, err)
            worst = max(worst, err)
    return worst


@pytest.mark.parametrize("H,W,S,u8", [(370, 1226, 32, True), (376, 1241, 16, True), (480, 640, 32, False), (1080, 1920, 8, True), (370, 1226, 6, False)])
def test_tolerance_batch_planes_vs_oracle(slam, syn, orc, monkeypatch, H, W, S, u8):
    """all 6 planes x 4 levels of the first, a middle and the last member of a tolerance-mode batch at the four BASELINE shapes; the
    kernel-selection threshold is lowered so that levels 1-2 take the tolerance kernels at these batch sizes too (the bench's S = 128 does)"""
    import torch
    monkeypatch.setenv("SLAMHIP_CK_MIN_MB", "1")
    fr = _frames(syn, H, W, S, seed=H + S)
    if u8:
        dev = torch.from_numpy(np.stack([np.ascontiguousarray(im.T) for im in fr])).cuda(); step = H * W
    else:
        dev = torch.from_numpy(np.stack([np.ascontiguousarray((im.astype(np.float64) / 255.0).T) for im in fr])).cuda(); step = H * W * 8
    torch.cuda.synchronize()
    ptrs = [dev.data_ptr() + s * step for s in range(S)]
    pb = slam.PyramidBatch((H, W), levels=3, S=S)
    pb.update_(ptrs, u8=u8, fast=True)
    pb.update_(ptrs, u8=u8, fast=True)                               # the cached graph replay
    for s in sorted({0, S // 2, S - 1}):
        ref = orc.pyr_build(np.asfortranarray(fr[s].astype(np.float64) / 255.0), 3, 1.0, 1)
        _check_planes(pb.pyramids[s], ref, 3, (H, W, S, s))
    # the exact mode on the same batch object afterwards is still bit-exact (the tolerance kernels leave nothing behind)
    pb.update_(ptrs, u8=u8)
    ref = orc.pyr_build(np.asfortranarray(fr[S - 1].astype(np.float64) / 255.0), 3, 1.0, 1)
    for l in range(4):
        for name in PLANES:
            assert np.array_equal(pb.pyramids[S - 1].plane(name, l), ref.plane(name, l)), (name, l)


def test_tolerance_batch_tracking_vs_oracle(slam, syn, orc, monkeypatch):
    """fb_tracking! between members of two tolerance-mode batches (consecutive frames of a stream) against the oracle on its exact
    planes, sequential summation order: positions <= 1e-6 px, status flips <= 0.5 %"""
    import torch
    monkeypatch.setenv("SLAMHIP_CK_MIN_MB", "1")
    H, W, S = 370, 1226, 8
    Ls, Rs, flows = syn.stereo_stream((H, W), 2, seed=21)
    def batch(img):
        u8 = np.asfortranarray(np.round(img * 255).astype(np.uint8))
        dev = torch.from_numpy(np.stack([np.ascontiguousarray(u8.T)] * S)).cuda()
        torch.cuda.synchronize()
        pb = slam.PyramidBatch((H, W), levels=3, S=S)
        pb.update_([dev.data_ptr() + s * H * W for s in range(S)], u8=True, fast=True)
        return pb, np.asfortranarray(u8.astype(np.float64) / 255.0)
    pa, fa = batch(Ls[0]); pc, fc = batch(Ls[1])
    kp = orc.detect(fa, np.zeros((0, 2)), max_points=1000).astype(float)
    got, st = slam.fb_tracking_(pa.pyramids[3], pc.pyramids[3], kp, window_size=9, pyramid_levels=3, max_distance=1.0)
    ra, rc = orc.pyr_build(fa, 3, 1.0, 1), orc.pyr_build(fc, 3, 1.0, 1)
    ro, rs = orc.fb_tracking(ra, rc, kp, sum_order=0)
    assert (st != rs).sum() <= max(1, len(kp) // 200), int((st != rs).sum())
    both = st & rs
    assert both.sum() > len(kp) // 2
    assert np.abs(got[both] - ro[both]).max() <= 1e-6


def test_tolerance_batch_s128_kitti_vs_oracle(slam, syn, orc):
    """The configuration bench.py quotes `tolerance_mode.value` on: 128 u8 frames of 370 x 1226 per build, DEFAULT kernel-selection
    thresholds (no SLAMHIP_CK_MIN_MB override), full and target-only builds; streams 0, 64 and 127 within 1e-11 of the oracle."""
    import torch
    H, W, S = 370, 1226, 128
    fr = _frames(syn, H, W, S, seed=17)
    dev = torch.from_numpy(np.stack([np.ascontiguousarray(im.T) for im in fr])).cuda()
    torch.cuda.synchronize()
    ptrs = [dev.data_ptr() + s * H * W for s in range(S)]
    pb = slam.PyramidBatch((H, W), levels=3, S=S)
    pb.update_(ptrs, u8=True, fast=True); pb.update_(ptrs, u8=True, fast=True)
    tg = slam.PyramidBatch((H, W), levels=3, S=S)
    tg.update_(ptrs, u8=True, fast=True, target_only=True); tg.update_(ptrs, u8=True, fast=True, target_only=True)
    for s in (0, 64, 127):
        ref = orc.pyr_build(np.asfortranarray(fr[s].astype(np.float64) / 255.0), 3, 1.0, 1)
        _check_planes(pb.pyramids[s], ref, 3, ("full", s))
        for l in range(4):                                           # what a tracking target needs: every layer + the finest level's planes
            g, r = tg.pyramids[s].plane("layers", l), ref.plane("layers", l)
            assert np.abs(g - r).max() / np.abs(r).max() <= TOL, ("target_only", s, l)
        for name in PLANES:
            g, r = tg.pyramids[s].plane(name, 0), ref.plane(name, 0)
            assert np.abs(g - r).max() / max(np.abs(r).max(), 1e-300) <= TOL, ("target_only", s, name)


def test_tolerance_kpset_match_vs_oracle(slam, syn, orc, monkeypatch):
    """slam_kpset_flow_match between two TOLERANCE-mode batches takes the contracted-arithmetic tracking kernel (lk.hip, TOL): against
    the oracle's optical_flow_matching! on its exact planes, sequential summation order -- positions <= 1e-6 px, at most 0.5 % of the
    keypoints differ in fate; SLAMHIP_NO_TOL_LK=1 (the exact kernel on the same planes) must agree to the same bar."""
    import torch
    monkeypatch.setenv("SLAMHIP_CK_MIN_MB", "1")
    H, W, S = 370, 1226, 4
    streams = [syn.stereo_stream((H, W), 2, seed=40 + s, step=(1.0 + 0.2 * s, -1.4)) for s in range(S)]
    u8 = lambda im: np.asfortranarray(np.round(im * 255).astype(np.uint8))
    def batch(k):
        fr = [u8(streams[s][0][k]) for s in range(S)]
        dev = torch.from_numpy(np.stack([np.ascontiguousarray(f.T) for f in fr])).cuda(); torch.cuda.synchronize()
        pb = slam.PyramidBatch((H, W), levels=3, S=S)
        pb.update_([dev.data_ptr() + s * H * W for s in range(S)], u8=True, fast=True)
        return pb, [np.asfortranarray(f.astype(np.float64) / 255.0) for f in fr]
    a, fa = batch(0); b, fb = batch(1)
    params = slam.Params(stereo=True, max_nb_keypoints=1000)
    rng = np.random.default_rng(3)
    kps = [orc.detect(fa[s], np.zeros((0, 2)), max_points=1000).astype(float) for s in range(S)]
    is3 = [rng.random(len(k)) < 0.6 for k in kps]
    shift = np.array([streams[s][2][1] for s in range(S)], dtype=np.float64) + rng.normal(0, 0.4, (S, 2))
    sp = slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=shift)
    ks = slam.KeypointSet(S, 1400)
    for s in range(S):
        ks.upload(s, kps[s], is3[s])
    ks.flow_match(a, b, params, sp, prior=2)
    n_diff = n_all = 0; worst = 0.0
    for s in range(S):
        got = ks.download(s)
        ra, rb = orc.pyr_build(fa[s], 3, 1.0, 1), orc.pyr_build(fb[s], 3, 1.0, 1)
        ref = orc.optical_flow_matching(ra, rb, kps[s], is3[s], kps[s] + shift[s], (H, W), sum_order=0)
        keep_ref = ~ref["removed"]
        n_all += len(kps[s])
        if len(got["yx"]) == keep_ref.sum():
            d = np.abs(got["yx"] - ref["new_pixels"][keep_ref]).max(axis=1)
            n_diff += int((d > 1e-6).sum()); worst = max(worst, float(d[d <= 1e-6].max()))
        else:
            n_diff += abs(len(got["yx"]) - int(keep_ref.sum())) + 1
    assert n_diff <= max(1, n_all // 200), (n_diff, n_all)
    assert worst <= 1e-6
    ks.close()
