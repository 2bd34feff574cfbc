"""GPU: the RANSAC seams on hostile inputs (coincident / collinear points, zero motion, huge and tiny magnitudes,
non-finite values): no hang, and still exactly the oracle's answer -- the solvers are the same arithmetic on both sides,
including their failure paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("kind", ["collinear", "coincident", "huge", "tiny", "nan", "random"])
def test_p3p_hostile_inputs_match_oracle(slam, orc, syn, kind):
    rng = np.random.default_rng(len(kind) * 131 + 7)
    sc = syn.p3p_scene(n=80, seed=3, iters=40)
    pts, px, pdn = sc["pts3d"].copy(), sc["px_xy"].copy(), sc["pdn"].copy()
    if kind == "collinear":
        pts[:, 1] = 2 * pts[:, 0]; pts[:, 2] = 5 + pts[:, 0]
    elif kind == "coincident":
        pts[:] = pts[0]; pdn[:] = pdn[0]
    elif kind == "huge":
        pts *= 1e150
    elif kind == "tiny":
        pts *= 1e-160
    elif kind == "nan":
        pts[::7] = np.nan; pdn[3::11] = np.inf
    else:
        pts = rng.normal(size=pts.shape) * 10; pdn = rng.normal(size=pdn.shape); px = rng.uniform(0, 1000, px.shape)
    ref = orc.p3p_ransac(pts, px, pdn, sc["K"], 3.0, sc["samples"])
    res = slam.p3p_ransac(pts, px, pdn, sc["K"], threshold=3.0, samples=sc["samples"], return_pose=True)
    if ref[0] == 0:
        assert res is None
    else:
        cnt, (KP, inl, err, Rt, bi) = res
        assert cnt == ref[0] and bi == ref[5] and _same(KP, ref[1]) and _same(Rt, ref[2]) and _same(inl, ref[3]) and _same(err, ref[4])


@pytest.mark.parametrize("kind", ["zero_motion", "coincident", "collinear", "huge", "nan", "random"])
def test_five_point_hostile_inputs_match_oracle(slam, orc, syn, kind):
    rng = np.random.default_rng(len(kind) * 131 + 7)
    sc = syn.five_point_scene(n=70, seed=5, iters=24)
    a, b, c, d = sc["px1"].copy(), sc["px2"].copy(), sc["pd1"].copy(), sc["pd2"].copy()
    if kind == "zero_motion":
        b[:] = a; d[:] = c
    elif kind == "coincident":
        a[:] = a[0]; b[:] = b[0]; c[:] = c[0]; d[:] = d[0]
    elif kind == "collinear":
        c[:, 1] = 0.5 * c[:, 0]; d[:, 1] = 0.5 * d[:, 0]; a[:, 1] = a[:, 0]; b[:, 1] = b[:, 0]
    elif kind == "huge":
        c *= 1e120; d *= 1e120
    elif kind == "nan":
        c[::9] = np.nan; d[4::13] = np.inf
    else:
        a = rng.uniform(0, 1200, a.shape); b = rng.uniform(0, 1200, b.shape); c = rng.normal(size=c.shape); d = rng.normal(size=d.shape)
    ref = orc.five_point_ransac(a, b, c, d, sc["K"], sc["K"], 3.0, sc["samples"])
    cnt, (E, P, inl, err, bi) = slam.five_point_ransac(a, b, c, d, sc["K"], sc["K"], max_repr_error=3.0, samples=sc["samples"], return_extra=True)
    assert cnt == ref[0] and bi == ref[5]
    assert _same(E, ref[1]) and _same(P, ref[2]) and _same(inl, ref[3]) and _same(err, ref[4])
