"""GPU: lock-stepped batches (slam_pyr_create_batch / update_batch_dev / flow_match_batch) give results
bit-identical to the single-image entry points -- batching only changes how many images share a launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PLANES = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")


@pytest.mark.parametrize("fast", [False, True])
def test_batch_pyramids_equal_single(slam, orc, texture, fast):
    import torch
    H, W, S = 101, 75, 3
    imgs = [texture(H, W, seed=s)[0][0] for s in range(S)] 
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
    torch.cuda.synchronize()
    batch = slam.PyramidBatch((H, W), levels=2, S=S)
    batch.update_([d.data_ptr() for d in dev], fast=fast)
    for s in range(S):
        single = slam.LKPyramid(shape=(H, W), levels=2)
        slam.update_(single, imgs[s], fast=fast)
        for l in range(3):
            for name in PLANES:
                assert np.array_equal(batch.pyramids[s].plane(name, l), single.plane(name, l)), (s, name, l)
        if not fast:
            ref = orc.pyr_build(imgs[s], 2, 1.0, 1)
            assert np.array_equal(batch.pyramids[s].plane("Iyx", 2), ref.plane("Iyx", 2))
    # members are ordinary handles: update one alone, the others keep their planes
    before = batch.pyramids[1].plane("Iy", 1)
    slam.update_(batch.pyramids[0], imgs[2], fast=fast)
    assert np.array_equal(batch.pyramids[0].plane("Ixx", 1), batch.pyramids[2].plane("Ixx", 1))
    assert np.array_equal(batch.pyramids[1].plane("Iy", 1), before)


def test_batch_flow_match_equals_per_stream(slam, orc, texture):
    import torch
    H, W, S = 120, 160, 3
    streams = [texture(H, W, seed=10 + s, step=(1.0 + 0.3 * s, -1.5)) for s in range(S)]
    a = slam.PyramidBatch((H, W), levels=3, S=S); b = slam.PyramidBatch((H, W), levels=3, S=S)
    d0 = [torch.from_numpy(np.ascontiguousarray(st[0][0].T)).cuda() for st in streams]
    d1 = [torch.from_numpy(np.ascontiguousarray(st[0][1].T)).cuda() for st in streams]
    torch.cuda.synchronize()
    a.update_([d.data_ptr() for d in d0]); b.update_([d.data_ptr() for d in d1])
    params = slam.Params()
    pts, idx, is3, proj = [], [], [], []
    ref_new, ref_st = [], []
    for s in range(S):
        kp = orc.detect(streams[s][0][0], np.zeros((0, 2)), max_points=150).astype(float)
        i3 = np.arange(len(kp)) % 3 != 0
        pr = kp + np.array(streams[s][2][1])
        pts.append(kp); idx.append(np.full(len(kp), s)); is3.append(i3); proj.append(pr)
        n1, s1 = slam.optical_flow_matching(a.pyramids[s], b.pyramids[s], kp, i3, pr, params)
        ref_new.append(n1); ref_st.append(s1)
    order = np.random.default_rng(0).permutation(sum(len(p) for p in pts))          # streams interleaved arbitrarily
    P = np.concatenate(pts)[order]; I = np.concatenate(idx)[order]; T = np.concatenate(is3)[order]; R = np.concatenate(proj)[order]
    new, st = slam.optical_flow_matching_batch(a, b, I, P, T, R, params)
    inv = np.argsort(order)
    assert np.array_equal(st[inv], np.concatenate(ref_st))
    assert np.array_equal(new[inv], np.concatenate(ref_new))
    with pytest.raises(slam.SlamHipError):
        slam.optical_flow_matching_batch(a, b, np.full(len(P), S), P, T, R, params)      # stream index out of range
    # the batched kernel against the CPU oracle's optical_flow_matching! directly
    for s in range(S):
        ra, rb = orc.pyr_build(streams[s][0][0], 3, 1.0, 1), orc.pyr_build(streams[s][0][1], 3, 1.0, 1)
        ref = orc.optical_flow_matching(ra, rb, pts[s], is3[s], proj[s], None, sum_order=1)
        sel = np.concatenate(idx) == s
        got_st, got_new = st[inv][sel], new[inv][sel]
        assert np.array_equal(got_st, ref["updated"])
        assert np.abs(got_new - ref["new_pixels"])[got_st].max() <= 1e-9


def test_checkpointed_row_kernel_is_bit_exact(slam, monkeypatch):
    """Bandwidth-bound batched launches switch both IIR passes to the checkpointed kernels (k_iir_rows_ck,
    k_iir_cols_ck: forward state checkpoints + recomputation instead of a stored forward plane).  Forced here on
    small batches: every plane must stay bit-identical to the single-image kernels (several block counts, incl.
    last blocks of 1 and of 32 samples, heights around the tile / block boundaries)."""
    import torch
    for (H, W) in ((70, 71), (33, 102), (130, 64 + 6 + 1), (97, 80), (64, 66), (65, 64), (69, 75), (100, 70),
                   (80, 62), (81, 63), (96, 124), (95, 125), (66, 187), (128, 61), (67, 8), (131, 250)):      # strip boundaries of the fused column kernel (62 own columns per workgroup)
        S = 2
        rng = np.random.default_rng(H * W)
        imgs = [np.asfortranarray(rng.random((H, W))) for _ in range(S)]
        dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
        torch.cuda.synchronize()
        monkeypatch.setenv("SLAMHIP_CK_MIN_MB", "0")
        batch = slam.PyramidBatch((H, W), levels=1, S=S)
        batch.update_([d.data_ptr() for d in dev])
        monkeypatch.delenv("SLAMHIP_CK_MIN_MB")
        for s in range(S):
            single = slam.LKPyramid(shape=(H, W), levels=1)
            slam.update_(single, imgs[s])
            for name in PLANES:
                assert np.array_equal(batch.pyramids[s].plane(name, 0), single.plane(name, 0)), (H, W, s, name)
            assert np.array_equal(batch.pyramids[s].plane("layers", 1), single.plane("layers", 1))


def test_batch_at_kitti_size_uses_checkpointed_kernels_and_stays_exact(slam, syn, orc):
    """8 images of 370 x 1226 per launch is above the bandwidth threshold (116 MB of plane data at level 0), so this
    batch runs k_iir_cols_ck / k_iir_rows_ck with their real block counts; members must equal single-image pyramids."""
    import torch
    H, W, S = 370, 1226, 8
    rng = np.random.default_rng(77)
    base = syn.texture_canvas(H, W, seed=5, margin=0)
    imgs = [np.asfortranarray(np.clip(base + 0.02 * rng.standard_normal((H, W)), 0, 1)) for _ in range(S)]
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
    torch.cuda.synchronize()
    batch = slam.PyramidBatch((H, W), levels=3, S=S)
    batch.update_([d.data_ptr() for d in dev])
    for s in (0, S - 1):
        single = slam.LKPyramid(shape=(H, W), levels=3)
        slam.update_(single, imgs[s])
        ref = orc.pyr_build(imgs[s], 3, 1.0, 1)                  # the batched kernels against the CPU oracle directly
        for l in range(4):
            for name in PLANES:
                assert np.array_equal(batch.pyramids[s].plane(name, l), single.plane(name, l)), (s, name, l)
                assert np.array_equal(batch.pyramids[s].plane(name, l), ref.plane(name, l)), ("oracle", s, name, l)
    # the same frames as a tracking-target-only batch (SLAM_PYR_TARGET_ONLY, fused / checkpointed kernels at this size): every layer
    # and the finest level's gradient / integral planes are the full build's, bit for bit
    tgt = slam.PyramidBatch((H, W), levels=3, S=S)
    tgt.update_([d.data_ptr() for d in dev], target_only=True)
    tgt.update_([d.data_ptr() for d in dev], target_only=True)       # (graph replay)
    for s in (0, S - 1):
        for l in range(4):
            assert np.array_equal(tgt.pyramids[s].plane("layers", l), batch.pyramids[s].plane("layers", l)), (s, l)
        for name in PLANES:
            assert np.array_equal(tgt.pyramids[s].plane(name, 0), batch.pyramids[s].plane(name, 0)), (s, name)


@pytest.mark.parametrize("shape", [(70, 71), (33, 102), (130, 135), (64, 64), (65, 129), (16, 200), (200, 17)])
def test_fused_integral_image_kernel_is_bit_exact(slam, shape, orc):
    """Batches of >= 8 images build the integral images with the one-pass kernel (k_cum_fused: 64 x 64 tiles, column
    sums then row sums with carries); planes must equal the two-pass single-image kernels bit for bit."""
    import torch
    H, W = shape
    S = 8
    rng = np.random.default_rng(H * 1000 + W)
    imgs = [np.asfortranarray(rng.random((H, W))) for _ in range(S)]
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
    torch.cuda.synchronize()
    batch = slam.PyramidBatch((H, W), levels=1, S=S)
    batch.update_([d.data_ptr() for d in dev])
    for s in (0, 3, S - 1):
        single = slam.LKPyramid(shape=(H, W), levels=1)
        slam.update_(single, imgs[s])
        for l in range(2):
            for name in PLANES:
                assert np.array_equal(batch.pyramids[s].plane(name, l), single.plane(name, l)), (shape, s, name, l)
    ref = orc.pyr_build(imgs[3], 1, 1.0, 1)
    for l in range(2):
        for name in PLANES:
            assert np.array_equal(batch.pyramids[3].plane(name, l), ref.plane(name, l)), ("oracle", shape, name, l)


def test_batch_u8_ingest_equals_float_ingest(slam, syn):
    """8-bit frames converted on the device (raw / 255, correctly rounded) == Float64 frames Gray{Float64}.(img)."""
    import torch
    H, W, S = 90, 121, 4
    rng = np.random.default_rng(4)
    u8 = [np.asfortranarray(rng.integers(0, 256, (H, W), dtype=np.uint8)) for _ in range(S)]
    d8 = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in u8]
    f64 = [torch.from_numpy(np.ascontiguousarray((im.astype(np.float64) / 255.0).T)).cuda() for im in u8]
    torch.cuda.synchronize()
    a = slam.PyramidBatch((H, W), levels=2, S=S); b = slam.PyramidBatch((H, W), levels=2, S=S)
    a.update_([d.data_ptr() for d in d8], u8=True)
    b.update_([d.data_ptr() for d in f64])
    for s in range(S):
        for l in range(3):
            for name in PLANES:
                assert np.array_equal(a.pyramids[s].plane(name, l), b.pyramids[s].plane(name, l)), (s, name, l)


def test_batch_at_fhd_size_stays_exact(slam):
    """BASELINE config 5 shape (1080 x 1920): above the 512 rows one workgroup of the one-pass integral kernel covers (levels 0 and 1 run
    it as chained row segments) and far above the bandwidth threshold (checkpointed IIR kernels at levels 0-2)."""
    import torch
    H, W, S = 1080, 1920, 8
    rng = np.random.default_rng(5)
    base = rng.random((H // 8 + 2, W // 8 + 2))
    imgs = []
    for s in range(S):
        up = np.kron(base, np.ones((8, 8)))[s:s + H, s:s + W]
        imgs.append(np.asfortranarray(0.5 * up + 0.5 * rng.random((H, W))))
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
    torch.cuda.synchronize()
    batch = slam.PyramidBatch((H, W), levels=3, S=S)
    batch.update_([d.data_ptr() for d in dev])
    for s in (0, S - 1):
        single = slam.LKPyramid(shape=(H, W), levels=3)
        slam.update_(single, imgs[s])
        for l in range(4):
            for name in PLANES:
                assert np.array_equal(batch.pyramids[s].plane(name, l), single.plane(name, l)), (s, name, l)


@pytest.mark.parametrize("H,W", [(513, 70), (577, 33), (777, 96), (1025, 130), (1100, 64), (2000, 40)])
def test_tall_planes_take_the_integral_kernel_in_chained_row_segments(slam, orc, H, W):
    """k_cum_fused beyond 512 rows: row segments of 256 rows, one workgroup each, the band chain carried through global memory -- every
    plane of every level against the oracle, twice (the hand-over flags must be clear again for the replay), incl. a one-band last segment"""
    import torch
    S = 8
    rng = np.random.default_rng(H + W)
    imgs = [np.asfortranarray(np.round(rng.random((H, W)) * 255).astype(np.uint8)) for _ in range(S)]
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
    torch.cuda.synchronize()
    levels = 3
    batch = slam.PyramidBatch((H, W), levels=levels, S=S)
    for rep in range(2):
        batch.update_([d.data_ptr() for d in dev], u8=True)
        for s in (0, S - 1):
            ref = orc.pyr_build(np.asfortranarray(imgs[s].astype(np.float64) / 255.0), levels, 1.0, 1)
            for l in range(levels + 1):
                for name in PLANES:
                    assert np.array_equal(batch.pyramids[s].plane(name, l), ref.plane(name, l)), (rep, s, name, l)


def test_two_tall_batches_built_at_once(slam):
    """two contexts build 1080-row batches at the same time: the chained row segments of k_cum_fused (a segment waits for the one above
    it, dispatched before it) of two launches share the chip; planes as a build alone gives them"""
    import threading, torch
    H, W, S = 1080, 640, 8
    rng = np.random.default_rng(11)
    imgs = [np.asfortranarray(np.round(rng.random((H, W)) * 255).astype(np.uint8)) for _ in range(S)]
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
    torch.cuda.synchronize()
    ref = slam.PyramidBatch((H, W), levels=3, S=S); ref.update_([d.data_ptr() for d in dev], u8=True)
    want = {(s, nm, l): ref.pyramids[s].plane(nm, l) for s in (0, S - 1) for nm in ("Iyy", "Iyx", "layers") for l in range(4)}
    out = {}
    def run(tag):
        ctx = slam.Context(0)
        b = slam.PyramidBatch((H, W), levels=3, S=S, ctx=ctx)
        for _ in range(8):
            b.update_([d.data_ptr() for d in dev], u8=True, ctx=ctx)
        out[tag] = {k: b.pyramids[k[0]].plane(k[1], k[2]) for k in want}
        ctx.close()
    th = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    [t.start() for t in th]; [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th)
    for t in range(2):
        for k, v in want.items():
            assert np.array_equal(out[t][k], v), (t, k)


def test_flow_match_batch_kept_is_the_compaction_of_flow_match_batch(slam, texture):
    import torch
    H, W, S = 120, 160, 3
    streams = [texture(H, W, seed=10 + s, step=(1.0 + 0.3 * s, -1.5)) for s in range(S)]
    a = slam.PyramidBatch((H, W), levels=3, S=S); b = slam.PyramidBatch((H, W), levels=3, S=S)
    d0 = [torch.from_numpy(np.ascontiguousarray(st[0][0].T)).cuda() for st in streams]
    d1 = [torch.from_numpy(np.ascontiguousarray(st[0][1].T)).cuda() for st in streams]
    torch.cuda.synchronize()
    a.update_([d.data_ptr() for d in d0]); b.update_([d.data_ptr() for d in d1])
    params = slam.Params()
    rng = np.random.default_rng(8)
    n = 400
    pts = np.stack([rng.uniform(1, H, n), rng.uniform(1, W, n)], axis=1)      # many of these fail (borders, flat areas)
    idx = np.sort(rng.integers(0, S, n)).astype(np.int32)
    is3 = rng.random(n) < 0.6
    proj = pts + np.array([1.0, -1.5])
    new, ok = slam.optical_flow_matching_batch(a, b, idx, pts, is3, proj, params)
    kp, k3, kidx, src = slam.optical_flow_matching_batch_kept(a, b, idx, pts, is3, proj, params)
    assert 0 < ok.sum() < n
    assert np.array_equal(src, np.flatnonzero(ok))
    assert np.array_equal(kp, new[ok]) and np.array_equal(k3, is3[ok]) and np.array_equal(kidx, idx[ok])
    _, ok2 = slam.optical_flow_matching_batch(a, b, idx, pts, is3, proj, params, status_only=True)
    assert np.array_equal(ok2, ok)


def test_event_markers_order_two_contexts(slam, texture):
    """slam_event_record / slam_ctx_wait_event: a consumer context waits for a marked point of a producer context's
    stream although more work has been enqueued behind it."""
    import torch
    H, W = 200, 300
    imgs = [texture(H, W, seed=s)[0][0] for s in range(3)]
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
    torch.cuda.synchronize()
    prod, cons = slam.Context(0), slam.Context(0)
    pyr = [slam.LKPyramid(shape=(H, W), levels=3, ctx=prod) for _ in range(3)]
    marks = []
    for k in range(3):                                   # three asynchronous builds, one marker each
        slam.update_(pyr[k], None, device_ptr=dev[k].data_ptr(), sync=False, ctx=prod)
        marks.append(prod.record())
    cons.wait_event(marks[0])
    got0 = pyr[0].plane("Iyx", 3, ctx=cons)           # copied on the consumer's stream, ordered by the marker only
    prod.synchronize()
    for k in range(3):
        ref = slam.LKPyramid(shape=(H, W), levels=3)
        slam.update_(ref, imgs[k])
        assert np.array_equal(pyr[k].plane("Iyx", 3), ref.plane("Iyx", 3))
        if k == 0:
            assert np.array_equal(got0, ref.plane("Iyx", 3))
    for m in marks:
        m.close()
    prod.close(); cons.close()
