"""GPU: slam_p3p_ransac vs the CPU oracle.  The solver uses only + - * / sqrt in the oracle's order, so the winning
hypothesis, the inlier mask and the pose are bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,iters,noise,outl", [(400, 256, 0.3, 0.2), (1000, 512, 0.5, 0.4), (37, 64, 0.0, 0.0), (5, 16, 0.2, 0.0),
                                                 (5000, 64, 0.4, 0.3)])   # > 4096 points: the select kernel's global-memory error path
def test_p3p_matches_oracle(slam, orc, syn, n, iters, noise, outl):
    sc = syn.p3p_scene(n=n, seed=n, noise_px=noise, outlier_frac=outl, iters=iters)
    ref = orc.p3p_ransac(sc["pts3d"], sc["px_xy"], sc["pdn"], sc["K"], 3.0, sc["samples"])
    res = slam.p3p_ransac(sc["pts3d"], sc["px_xy"], sc["pdn"], sc["K"], threshold=3.0, samples=sc["samples"], return_pose=True)
    assert res is not None
    cnt, (KP, inl, err, Rt, bi) = res
    assert cnt == ref[0] and bi == ref[5]
    assert np.array_equal(inl, ref[3])
    assert np.array_equal(Rt, ref[2]) and np.array_equal(KP, ref[1])
    assert err == ref[4]
    assert not inl[sc["gross"]].any()
    assert np.abs(Rt - sc["Rt_gt"]).max() < 0.2


def test_p3p_feeds_pnp_refinement(slam, syn):
    """compute_pose! (front_end.jl:164-217): RANSAC pose -> inliers -> pnp_bundle_adjustment."""
    sc = syn.p3p_scene(n=600, seed=9, noise_px=0.4, outlier_frac=0.3)
    cnt, (KP, inl, err, Rt, bi) = slam.p3p_ransac(sc["pts3d"], sc["px_xy"], sc["pdn"], sc["K"], threshold=3.0,
                                                    samples=sc["samples"], return_pose=True)
    K = sc["K"]
    assert np.allclose(np.linalg.inv(K) @ KP, Rt, atol=1e-9)                       # iK * KP, front_end.jl:182
    pose = np.eye(4); pose[:3] = Rt
    cam = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    new_T, e0, e1, outliers, n_out = slam.pnp_bundle_adjustment(cam, pose, sc["px_xy"][inl][:, ::-1], sc["pts3d"][inl],
                                                                repr_eps=3.0)
    assert e1 <= e0 and cnt - n_out >= 5
    assert np.abs(new_T[:3] - sc["Rt_gt"]).max() < np.abs(Rt - sc["Rt_gt"]).max() + 1e-9
    assert np.abs(new_T[:3, 3] - sc["Rt_gt"][:, 3]).max() < 0.05


def test_p3p_no_model_and_small_inputs(slam, syn):
    sc = syn.p3p_scene(n=50, seed=4)
    bad = np.array([[0, 0, 1], [5, 60, 2], [-1, 2, 3]], dtype=np.int32)
    assert slam.p3p_ransac(sc["pts3d"], sc["px_xy"], sc["pdn"], sc["K"], threshold=3.0, samples=bad) is None
    assert slam.p3p_ransac(sc["pts3d"][:2], sc["px_xy"][:2], sc["pdn"][:2], sc["K"], threshold=3.0) is None
    assert slam.p3p_ransac(np.zeros((0, 3)), np.zeros((0, 2)), np.zeros((0, 3)), sc["K"]) is None
    with pytest.raises(ValueError):
        slam.p3p_ransac(sc["pts3d"], sc["px_xy"][:10], sc["pdn"], sc["K"])


def test_p3p_default_sampler_is_reproducible(slam, syn):
    sc = syn.p3p_scene(n=300, seed=6)
    a = slam.p3p_ransac(sc["pts3d"], sc["px_xy"], sc["pdn"], sc["K"], threshold=3.0, iterations=128, seed=5, return_pose=True)
    b = slam.p3p_ransac(sc["pts3d"], sc["px_xy"], sc["pdn"], sc["K"], threshold=3.0, iterations=128, seed=5, return_pose=True)
    assert a[0] == b[0] and np.array_equal(a[1][3], b[1][3]) and a[0] > 200
