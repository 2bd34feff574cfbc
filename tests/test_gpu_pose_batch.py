"""GPU: the pose seams for S lock-stepped streams (slam_p3p_ransac_batch, slam_pnp_ba_batch,
slam_five_point_ransac_batch): every problem of a batch gives exactly what the single-call entry point gives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p3p_scenes(syn, sizes, iters):
    return [syn.p3p_scene(n=n, seed=10 + k, noise_px=0.3, outlier_frac=0.2, iters=iters) for k, n in enumerate(sizes)]


def test_p3p_batch_equals_single_calls(slam, syn):
    sizes = [300, 5, 1000, 4500, 64]                       # ragged, incl. > 4096 (global error path)
    sc = _p3p_scenes(syn, sizes, 48)
    Ks = [s["K"] * (1 + 0.01 * k) for k, s in enumerate(sc)]          # per-stream intrinsics
    for k, s in enumerate(sc):                                         # keep the data consistent with the scaled K
        s["K"] = Ks[k]
    res = slam.p3p_ransac_batch([s["pts3d"] for s in sc], [s["px_xy"] for s in sc], [s["pdn"] for s in sc], Ks, threshold=3.0,
                                samples=[s["samples"] for s in sc])
    assert len(res) == len(sc)
    for s, r in zip(sc, res):
        one = slam.p3p_ransac(s["pts3d"], s["px_xy"], s["pdn"], s["K"], threshold=3.0, samples=s["samples"], return_pose=True)
        assert (r is None) == (one is None)
        if r is not None:
            assert r[0] == one[0] and r[1][4] == one[1][4] and r[1][2] == one[1][2]
            assert np.array_equal(r[1][0], one[1][0]) and np.array_equal(r[1][1], one[1][1]) and np.array_equal(r[1][3], one[1][3])


def test_p3p_batch_with_empty_and_tiny_problems(slam, syn):
    sc = _p3p_scenes(syn, [200, 200], 32)
    pts = [sc[0]["pts3d"], np.zeros((0, 3)), sc[1]["pts3d"][:2], sc[1]["pts3d"]]
    px = [sc[0]["px_xy"], np.zeros((0, 2)), sc[1]["px_xy"][:2], sc[1]["px_xy"]]
    bd = [sc[0]["pdn"], np.zeros((0, 3)), sc[1]["pdn"][:2], sc[1]["pdn"]]
    sm = [sc[0]["samples"], sc[0]["samples"], sc[1]["samples"], sc[1]["samples"]]
    res = slam.p3p_ransac_batch(pts, px, bd, sc[0]["K"], threshold=3.0, samples=sm)
    assert res[1] is None and res[2] is None and res[0] is not None and res[3] is not None
    one = slam.p3p_ransac(sc[1]["pts3d"], sc[1]["px_xy"], sc[1]["pdn"], sc[1]["K"], threshold=3.0, samples=sc[1]["samples"], return_pose=True)
    assert res[3][0] == one[0] and np.array_equal(res[3][1][3], one[1][3])
    assert slam.p3p_ransac_batch([], [], [], sc[0]["K"]) == []
    with pytest.raises(ValueError):
        slam.p3p_ransac_batch([sc[0]["pts3d"]], [sc[0]["px_xy"][:5]], [sc[0]["pdn"]], sc[0]["K"])


def test_pnp_batch_equals_single_calls(slam, syn):
    sc = [syn.pnp_scene(n=n, seed=k) for k, n in enumerate([300, 40, 1200, 6])]
    res = slam.pnp_bundle_adjustment_batch([s["cam"] for s in sc], [s["pose0"] for s in sc], [s["pixels_yx"] for s in sc],
                                           [s["points"] for s in sc], repr_eps=3.0)
    for s, r in zip(sc, res):
        one = slam.pnp_bundle_adjustment(s["cam"], s["pose0"], s["pixels_yx"], s["points"], repr_eps=3.0)
        assert np.array_equal(r[0], one[0]) and r[1] == one[1] and r[2] == one[2] and r[4] == one[4]
        assert np.array_equal(r[3], one[3])
    assert slam.pnp_bundle_adjustment_batch(sc[0]["cam"], [], [], []) == []


def test_five_point_batch_equals_single_calls(slam, syn):
    sizes = [300, 8, 1000, 4400, 3]
    sc = [syn.five_point_scene(n=n, seed=20 + k, noise_px=0.3, outlier_frac=0.2, iters=24) for k, n in enumerate(sizes)]
    sc[4]["samples"] = np.full((24, 5), -1, dtype=np.int32)               # three correspondences: nothing to sample
    res = slam.five_point_ransac_batch([s["px1"] for s in sc], [s["px2"] for s in sc], [s["pd1"] for s in sc], [s["pd2"] for s in sc],
                                       sc[0]["K"], sc[0]["K"], max_repr_error=3.0, samples=[s["samples"] for s in sc])
    for s, r in zip(sc, res):
        cnt, (E, P, inl, err, bi) = slam.five_point_ransac(s["px1"], s["px2"], s["pd1"], s["pd2"], s["K"], s["K"], max_repr_error=3.0,
                                                            samples=s["samples"], return_extra=True)
        assert r[0] == cnt and r[1][4] == bi and r[1][3] == err
        assert np.array_equal(r[1][0], E) and np.array_equal(r[1][1], P) and np.array_equal(r[1][2], inl)
    assert res[4][0] == 0                                                  # fewer than five correspondences


def test_compute_pose_chain_for_a_batch(slam, syn):
    """compute_pose! for S streams: one P3P launch set, then one PnP launch on each stream's inliers."""
    sc = _p3p_scenes(syn, [500, 700, 350], 96)
    res = slam.p3p_ransac_batch([s["pts3d"] for s in sc], [s["px_xy"] for s in sc], [s["pdn"] for s in sc], sc[0]["K"], threshold=3.0,
                                samples=[s["samples"] for s in sc])
    K = sc[0]["K"]; cam = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    poses, pix, pts = [], [], []
    for s, r in zip(sc, res):
        T = np.eye(4); T[:3] = r[1][3]
        inl = r[1][1]
        poses.append(T); pix.append(s["px_xy"][inl][:, ::-1]); pts.append(s["pts3d"][inl])
    ref = slam.pnp_bundle_adjustment_batch(cam, poses, pix, pts, repr_eps=3.0)
    for s, r in zip(sc, ref):
        assert r[2] <= r[1]
        assert np.abs(r[0][:3, 3] - s["Rt_gt"][:, 3]).max() < 0.05
