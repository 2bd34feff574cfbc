"""GPU: examples/kitti_frontend.py on a synthetic KITTI-shaped sequence: 8-bit PNG frames -> slam_pyr_update_u8 ->
tracking -> key-frame detection -> stereo matching -> triangulation, end to end through the host mirror."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example():
    spec = importlib.util.spec_from_file_location("kitti_frontend", os.path.join(ROOT, "examples", "kitti_frontend.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_frontend_over_synthetic_sequence(slam, syn, tmp_path):
    from slam_jl_amd import kitti
    disparity, baseline = 6.0, 0.54
    L, R, flows = syn.stereo_stream((140, 260), 7, seed=3, step=(0.7, -1.2), disparity=disparity)
    kitti.write_synthetic_sequence(str(tmp_path), "05", L, R, syn.KITTI_CAM, baseline)
    ds = slam.KittyDataset(str(tmp_path), "05", stereo=True)
    st = _example().run(ds, 7, kf_every=3, max_keypoints=120)
    assert [r["keyframe"] for r in st] == [True, False, False, True, False, False, True]
    assert st[0]["detected"] > 20 and st[0]["tracked"] == 0
    alive = st[0]["detected"]
    for r in st[1:3]:
        assert r["tracked"] >= 0.8 * alive                      # a rigid 1.4 px/frame shift: nearly everything survives
        alive = r["tracked"]
    # stereo: the right image is the left one shifted by `disparity` px -> depth = fx * baseline / disparity
    want = syn.KITTI_CAM[0] * baseline / disparity
    assert st[0]["stereo"] >= 0.7 * st[0]["detected"]
    assert abs(st[0]["depth_median"] - want) / want < 0.02
    # the 8-bit path is the same arithmetic as feeding bytes / 255 as Float64
    a, _ = ds[2]
    p8 = slam.LKPyramid(shape=a.shape, levels=3); slam.update_(p8, a)
    pf = slam.LKPyramid(shape=a.shape, levels=3); slam.update_(pf, a.astype(np.float64) / 255)
    for name in ("layers", "Iy", "Ixx", "Iyx"):
        assert np.array_equal(p8.plane(name, 2), pf.plane(name, 2))
