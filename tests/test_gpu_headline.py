"""GPU: the launch configuration of the bench headline itself against the CPU oracle -- not a scaled-down cousin of it.

bench.py's `value` runs on: 64 8-bit frames of 370 x 1226 per build through the fused u8 ingest of k_cols_fused (selected only
when a level carries >= 40 MB of plane data), the checkpointed / fused kernels at levels 0-2, tracking-target-only right builds
from u8, and slam_kpset_* on ~1000-keypoint lists per stream.  Each of these is selected by size, so each is compared with the
oracle here AT that size (pyramid.jl:81-137, map_manager.jl:451-564, extractor.jl:63-95)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PLANES = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")
H, W = 370, 1226


def _u8_frames(syn, S, seed):
    rng = np.random.default_rng(seed)
    base = syn.texture_canvas(H, W, seed=seed, margin=0)
    out = []
    for _ in range(S):
        im = np.clip(base + 0.03 * rng.standard_normal((H, W)), 0, 1)
        out.append(np.asfortranarray(np.round(im * 255).astype(np.uint8)))
    return out


def test_batch_u8_s64_kitti_vs_oracle(slam, syn, orc):
    """64 u8 frames of 370 x 1226 -> slam_pyr_update_batch_u8_dev: all 6 planes x 4 levels equal to the oracle's build of
    frame / 255.0 for streams 0, 31, 63; the same frames as a target-only batch: every layer + the finest level's planes."""
    import torch
    S = 64
    u8 = _u8_frames(syn, S, seed=11)
    dev = torch.from_numpy(np.stack([np.ascontiguousarray(im.T) for im in u8])).cuda()      # (S, W, H) = column-major H x W each
    torch.cuda.synchronize()
    ptrs = [dev.data_ptr() + s * H * W for s in range(S)]
    full = slam.PyramidBatch((H, W), levels=3, S=S)
    full.update_(ptrs, u8=True)
    full.update_(ptrs, u8=True)                                   # second call = the cached hipGraph replay the bench times
    tgt = slam.PyramidBatch((H, W), levels=3, S=S)
    tgt.update_(ptrs, u8=True, target_only=True)
    tgt.update_(ptrs, u8=True, target_only=True)
    for s in (0, 31, 63):
        ref = orc.pyr_build(np.asfortranarray(u8[s].astype(np.float64) / 255.0), 3, 1.0, 1)
        for l in range(4):
            for name in PLANES:
                assert np.array_equal(full.pyramids[s].plane(name, l), ref.plane(name, l)), ("full", s, name, l)
            assert np.array_equal(tgt.pyramids[s].plane("layers", l), ref.plane("layers", l)), ("target_only", s, l)
        for name in PLANES:
            assert np.array_equal(tgt.pyramids[s].plane(name, 0), ref.plane(name, 0)), ("target_only", s, name)


def test_batch_u8_s128_kitti_vs_oracle(slam, syn, orc):
    """128 u8 frames of 370 x 1226 per build -- bench.py's default batch since the end of round 3 (the largest a batch can be:
    slam_pyr_create_batch's limit): full and target-only builds, streams 0, 64 and 127 against the oracle, all planes and levels."""
    import torch
    S = 128
    u8 = _u8_frames(syn, S, seed=13)
    dev = torch.from_numpy(np.stack([np.ascontiguousarray(im.T) for im in u8])).cuda()
    torch.cuda.synchronize()
    ptrs = [dev.data_ptr() + s * H * W for s in range(S)]
    full = slam.PyramidBatch((H, W), levels=3, S=S)
    full.update_(ptrs, u8=True); full.update_(ptrs, u8=True)
    tgt = slam.PyramidBatch((H, W), levels=3, S=S)
    tgt.update_(ptrs, u8=True, target_only=True); tgt.update_(ptrs, u8=True, target_only=True)
    for s in (0, 64, 127):
        ref = orc.pyr_build(np.asfortranarray(u8[s].astype(np.float64) / 255.0), 3, 1.0, 1)
        for l in range(4):
            for name in PLANES:
                assert np.array_equal(full.pyramids[s].plane(name, l), ref.plane(name, l)), ("full", s, name, l)
            assert np.array_equal(tgt.pyramids[s].plane("layers", l), ref.plane("layers", l)), ("target_only", s, l)
        for name in PLANES:
            assert np.array_equal(tgt.pyramids[s].plane(name, 0), ref.plane(name, 0)), ("target_only", s, name)


def test_batch_f64_s48_kitti_vs_oracle(slam, syn, orc):
    """S = 48 Float64 frames (the streams_sweep leg; levels 1 and 2 cross the 40 MB kernel-selection threshold between S = 16 and
    S = 48): streams 0 and 47 against the oracle."""
    import torch
    S = 48
    u8 = _u8_frames(syn, S, seed=12)
    f64 = [np.asfortranarray(im.astype(np.float64) / 255.0) for im in u8]
    dev = torch.from_numpy(np.stack([np.ascontiguousarray(im.T) for im in f64])).cuda()
    torch.cuda.synchronize()
    ptrs = [dev.data_ptr() + s * H * W * 8 for s in range(S)]
    b = slam.PyramidBatch((H, W), levels=3, S=S)
    b.update_(ptrs)
    for s in (0, 47):
        ref = orc.pyr_build(f64[s], 3, 1.0, 1)
        for l in range(4):
            for name in PLANES:
                assert np.array_equal(b.pyramids[s].plane(name, l), ref.plane(name, l)), (s, name, l)


def test_kpset_keyframe_cycle_at_kitti_size_vs_oracle(slam, syn, orc, texture):
    """One key-frame cycle of the headline loop -- temporal match, detect + merge, stereo match -- on device-resident lists at
    370 x 1226, S = 16, 1000 keypoints per stream, u8 frames; streams 0, 7, 15 are replayed through the oracle's
    optical_flow_matching! and detect (the other streams share every launch)."""
    import torch
    S = 16
    params = slam.Params(stereo=True, max_nb_keypoints=1000)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    e = slam.Extractor.from_params(params, cam)
    disparity = 12.4
    streams = [texture(H, W, n=2, seed=50 + s, step=(1.3 - 0.05 * s, -2.1 + 0.1 * s), disparity=disparity) for s in range(S)]
    q = lambda im: np.asfortranarray(np.round(im * 255).astype(np.uint8))
    l0 = [q(st[0][0]) for st in streams]; l1 = [q(st[0][1]) for st in streams]; r1 = [q(st[1][1]) for st in streams]
    f = lambda im: np.asfortranarray(im.astype(np.float64) / 255.0)

    def batch(frames, **kw):
        d = torch.from_numpy(np.stack([np.ascontiguousarray(im.T) for im in frames])).cuda()
        torch.cuda.synchronize()
        b = slam.PyramidBatch((H, W), levels=3, S=S)
        b.update_([d.data_ptr() + s * H * W for s in range(S)], u8=True, **kw)
        return b, d

    a, da = batch(l0); b, db = batch(l1); r, dr = batch(r1, target_only=True)
    ncell = e.grid_resolution[0] * e.grid_resolution[1]
    cap = e.max_points + ncell + 8
    ks = slam.KeypointSet(S, cap)
    rng = np.random.default_rng(3)
    kps, is3 = [], []
    for s in range(S):
        k = orc.detect(f(l0[s]), np.zeros((0, 2)), max_points=1000).astype(float)
        k = k[rng.random(len(k)) >= 0.15]                       # the map culled some: detection has work at the key-frame
        kps.append(k); is3.append(rng.random(len(k)) < 0.6)
        ks.upload(s, k, is3[s])
    shift = np.array([st[2][1] for st in streams]) + rng.normal(0, 0.5, (S, 2))        # motion-model prior, ~0.5 px off
    ks.flow_match(a, b, params, slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=shift), prior=2)
    cnt = ks.counts()
    check = (0, 7, 15)
    lists = {}
    for s in check:
        ra, rb = orc.pyr_build(f(l0[s]), 3, 1.0, 1), orc.pyr_build(f(l1[s]), 3, 1.0, 1)
        ref = orc.optical_flow_matching(ra, rb, kps[s], is3[s], kps[s] + shift[s], (H, W), sum_order=1, threads=4)
        keep = ~ref["removed"]
        got = ks.download(s)
        assert cnt[s] == keep.sum() == len(got["yx"]), s
        assert np.array_equal(got["is_3d"], is3[s][keep]), s
        assert np.abs(got["yx"] - ref["new_pixels"][keep]).max() <= 1e-9, s
        assert keep.mean() > 0.9
        lists[s] = (got["yx"], got["is_3d"])
    ks.detect(e, b)
    cnt2 = ks.counts()
    for s in check:
        cur, t3 = lists[s]
        fresh = orc.detect(f(l1[s]), cur, max_points=1000).astype(float)
        got = ks.download(s)
        assert cnt2[s] == len(cur) + len(fresh), s
        assert np.array_equal(got["yx"][:len(cur)], cur) and np.array_equal(got["yx"][len(cur):], fresh), s
        assert len(fresh) > 50
        lists[s] = (got["yx"], np.concatenate([t3, np.zeros(len(fresh), bool)]))
    ks.stereo_match(b, r, params, slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=np.tile([0.0, -disparity], (S, 1))), prior=2)
    for s in check:
        kp, t3 = lists[s]
        rb, rr = orc.pyr_build(f(l1[s]), 3, 1.0, 1), orc.pyr_build(f(r1[s]), 3, 1.0, 1)
        ref = orc.optical_flow_matching(rb, rr, kp, t3, kp + np.array([0.0, -disparity]), (H, W), stereo=True, undistorted_left=kp,
                                        right_cam=syn.KITTI_CAM, sum_order=1, threads=4)
        got = ks.download(s)
        keep = ~ref["removed"]
        assert np.array_equal(got["yx"], kp[keep]), s
        assert np.array_equal(got["has_stereo"], ref["updated"][keep]), s
        up = got["has_stereo"]
        assert np.abs(got["stereo_yx"][up] - ref["new_pixels"][keep][up]).max() <= 1e-9, s
        assert up.mean() > 0.8
    ks.close()


@pytest.mark.parametrize("shape,S", [((376, 1241), 64), ((480, 640), 64), ((1080, 1920), 16), ((376, 1241), 128), ((480, 640), 128), ((1080, 1920), 32)])
def test_config_shapes_u8_batches_vs_oracle(slam, syn, orc, shape, S):
    """The other BASELINE shapes at the batch sizes bench.py's `configs` legs run them with (kitti00_2000 S = 128, euroc_mono S = 128,
    fhd_4000 S = 32; until the last day of round 3: 64 / 64 / 16; 8-bit ingest): first and last stream against the oracle, all planes and levels.  (1080 rows: above the 512-row
    limit of the one-pass integral kernel -> k_cum_cols + k_cum_rows; odd 1241 -> 621 columns: general bilinear resize.)"""
    import torch
    Hc, Wc = shape
    rng = np.random.default_rng(Hc)
    base = syn.texture_canvas(Hc, Wc, seed=3, margin=0)
    u8 = [np.asfortranarray(np.round(np.clip(base + 0.03 * rng.standard_normal((Hc, Wc)), 0, 1) * 255).astype(np.uint8)) for _ in range(2)]
    frames = [u8[s % 2] if s not in (0, S - 1) else np.asfortranarray(np.roll(u8[s % 2], s + 1, axis=1)) for s in range(S)]
    dev = torch.from_numpy(np.stack([np.ascontiguousarray(im.T) for im in frames])).cuda()
    torch.cuda.synchronize()
    b = slam.PyramidBatch((Hc, Wc), levels=3, S=S)
    ptrs = [dev.data_ptr() + s * Hc * Wc for s in range(S)]
    b.update_(ptrs, u8=True); b.update_(ptrs, u8=True)
    for s in (0, S - 1):
        ref = orc.pyr_build(np.asfortranarray(frames[s].astype(np.float64) / 255.0), 3, 1.0, 1)
        for l in range(4):
            for name in PLANES:
                assert np.array_equal(b.pyramids[s].plane(name, l), ref.plane(name, l)), (shape, s, name, l)
