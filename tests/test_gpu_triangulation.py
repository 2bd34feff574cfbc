"""GPU: slam_triangulate vs the CPU oracle (same operation order: positions to 1e-12 relative, status identical)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("temporal", [False, True])
def test_triangulate_matches_oracle(slam, orc, syn, temporal):
    s = syn.triangulation_scene(n=1500, seed=11, noise_px=0.4, n_behind=40, n_gross=60, temporal=temporal)
    P1, P2 = slam.projection_matrices(s["cam"], s["cam"], s["T21"])
    par = None
    if temporal:
        par = np.random.default_rng(2).uniform(0, 50, 1500)
    ref_xyz, ref_st = orc.triangulate(P1, P2, s["T21"], s["cam"], s["cam"], s["px1"], s["px2"], 3.0, parallax=par)
    xyz, st = slam.triangulate(s["cam"], s["cam"], s["T21"], s["px1"], s["px2"], 3.0, parallax=par)
    assert np.array_equal(st, ref_st)
    assert 0.5 < st.mean() < 1.0
    scale = np.abs(ref_xyz).max(axis=1, keepdims=True)
    assert np.max(np.abs(xyz - ref_xyz) / scale) < 1e-12
    ok = st & ~np.isin(np.arange(1500), np.concatenate([s["behind"], s["gross"]]))
    err = np.linalg.norm(xyz[ok] - s["xyz"][ok], axis=1) / s["xyz"][ok, 2]
    assert np.median(err) < 0.05


def test_triangulate_empty_and_single(slam, syn):
    s = syn.triangulation_scene(n=1, seed=1)
    xyz, st = slam.triangulate(s["cam"], s["cam"], s["T21"], np.zeros((0, 2)), np.zeros((0, 2)), 3.0)
    assert xyz.shape == (0, 3) and st.shape == (0,)
    xyz, st = slam.triangulate(s["cam"], s["cam"], s["T21"], s["px1"], s["px2"], 3.0)
    assert st[0] and np.allclose(xyz[0], s["xyz"][0], rtol=1e-8)
