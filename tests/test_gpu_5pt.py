"""GPU: slam_five_point_ransac vs the CPU oracle.  Solver, pose recovery and triangulation use only + - * / sqrt in
the oracle's order: winner, inlier mask, E, [R | t] and the summed error are bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,iters,noise,outl", [(300, 64, 0.3, 0.25), (1000, 96, 0.5, 0.4), (40, 32, 0.0, 0.0), (8, 16, 0.2, 0.0),
                                                 (4500, 12, 0.4, 0.3)])    # > 4096 correspondences: global-memory error path
def test_five_point_matches_oracle(slam, orc, syn, n, iters, noise, outl):
    sc = syn.five_point_scene(n=n, seed=n, noise_px=noise, outlier_frac=outl, iters=iters)
    ref = orc.five_point_ransac(sc["px1"], sc["px2"], sc["pd1"], sc["pd2"], sc["K"], sc["K"], 3.0, sc["samples"])
    cnt, (E, P, inl, err, bi) = slam.five_point_ransac(sc["px1"], sc["px2"], sc["pd1"], sc["pd2"], sc["K"], sc["K"],
                                                        max_repr_error=3.0, samples=sc["samples"], return_extra=True)
    assert cnt == ref[0] and bi == ref[5] and cnt >= 5
    assert np.array_equal(inl, ref[3])
    assert np.array_equal(P, ref[2]) and np.array_equal(E, ref[1])
    assert err == ref[4]
    assert np.abs(P[:, :3] - sc["Rt_gt"][:, :3]).max() < 0.05


def test_five_point_no_model_and_small_inputs(slam, syn):
    sc = syn.five_point_scene(n=50, seed=4)
    bad = np.array([[0, 0, 1, 2, 3], [5, 60, 2, 1, 0], [-1, 2, 3, 4, 5]], dtype=np.int32)
    cnt, (E, P, inl, err) = slam.five_point_ransac(sc["px1"], sc["px2"], sc["pd1"], sc["pd2"], sc["K"], sc["K"], 3.0, samples=bad)
    assert cnt == 0 and not inl.any() and not P.any()
    cnt, model = slam.five_point_ransac(sc["px1"][:4], sc["px2"][:4], sc["pd1"][:4], sc["pd2"][:4], sc["K"], sc["K"], 3.0)
    assert cnt == 0
    z = np.zeros((0, 2))
    cnt, model = slam.five_point_ransac(z, z, z, z, sc["K"], sc["K"])
    assert cnt == 0 and model[2].shape == (0,)
    with pytest.raises(ValueError):
        slam.five_point_ransac(sc["px1"], sc["px2"][:10], sc["pd1"], sc["pd2"], sc["K"], sc["K"])


def test_five_point_default_sampler(slam, syn):
    sc = syn.five_point_scene(n=400, seed=6, noise_px=0.3, outlier_frac=0.2)
    a = slam.five_point_ransac(sc["px1"], sc["px2"], sc["pd1"], sc["pd2"], sc["K"], sc["K"], 3.0, iterations=64, seed=5)
    b = slam.five_point_ransac(sc["px1"], sc["px2"], sc["pd1"], sc["pd2"], sc["K"], sc["K"], 3.0, iterations=64, seed=5)
    assert a[0] == b[0] and np.array_equal(a[1][1], b[1][1]) and a[0] > 250
    # compute_pose_5pt! rescales the unit translation with the motion-model baseline (front_end.jl:321-329)
    assert abs(np.linalg.norm(a[1][1][:, 3]) - 1.0) < 1e-12
