"""slam_detect_batch: one launch for the S streams of a pyramid batch == S slam_detect_pyr calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S", [1, 3, 8, 40])
def test_detect_batch_equals_per_stream(slam, syn, orc, S):
    H, W = 188, 620
    imgs = [np.asfortranarray(syn.texture_canvas(H, W, seed=50 + s, margin=0)) for s in range(S)]
    params = slam.Params(stereo=True, max_nb_keypoints=400)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    e = slam.Extractor.from_params(params, cam)
    import torch
    dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
    torch.cuda.synchronize()
    batch = slam.PyramidBatch((H, W), levels=3, S=S)
    batch.update_([d.data_ptr() for d in dev])
    rng = np.random.default_rng(3)
    cur, sid = [], []
    for s in range(S):
        # stream 0: no current points; last stream (S > 1): already full -> nothing detected; others: a partial set
        n = 0 if s == 0 else (params.max_nb_keypoints + 5 if (s == S - 1 and S > 1) else int(rng.integers(20, 300)))
        pts = np.stack([rng.uniform(1, H, n), rng.uniform(1, W, n)], axis=1)
        cur.append(pts); sid.append(np.full(n, s, dtype=np.int32))
    cur_all, sid_all = np.concatenate(cur), np.concatenate(sid)
    kp, ksid = slam.detect_batch(e, batch, cur_all, sid_all)
    total = 0
    for s in range(S):
        ref = slam.detect(e, batch.pyramids[s], cur[s])
        got = kp[ksid == s]
        assert np.array_equal(got, ref), f"stream {s}"
        if s in (0, 1, S - 2):                                   # the batched kernel against the CPU oracle directly
            oref = orc.detect(imgs[s], cur[s], max_points=params.max_nb_keypoints)
            assert np.array_equal(got, oref), f"oracle, stream {s}"
        total += len(ref)
    assert total == len(kp) and total > 0
    if S > 1:
        assert not np.any(ksid == S - 1)


def test_detect_batch_rejects_ungrouped(slam, syn):
    H, W = 64, 96
    params = slam.Params(stereo=True, max_nb_keypoints=50)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    e = slam.Extractor.from_params(params, cam)
    batch = slam.PyramidBatch((H, W), levels=1, S=2)
    with pytest.raises(ValueError):
        slam.detect_batch(e, batch, np.ones((3, 2)), np.array([1, 0, 1]))
