"""GPU: examples/device_frontend.py -- the per-frame front-end on device-resident keypoint lists (tracking with pose priors,
five-point filter, P3P + PnP pose, key-frame detection / stereo matching / triangulation) recovers the camera motion of a rigid
synthetic scene: a fronto-parallel plane at Z = fx b / d seen by cameras translating parallel to it."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example():
    spec = importlib.util.spec_from_file_location("device_frontend", os.path.join(ROOT, "examples", "device_frontend.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_device_frontend_recovers_the_camera_motion(slam, syn):
    ex = _example()
    S, n_frames, disparity, baseline = 3, 10, 8.0, 0.54
    cam = syn.KITTI_CAM
    lefts, rights, offs = ex.synthetic_scene(S, n_frames, shape=(200, 320), disparity=disparity, seed=5)
    out, n3 = ex.run(lefts, rights, cam, baseline, kf_every=4, max_keypoints=300, seed=9)
    Z = cam[0] * baseline / disparity
    assert [r["keyframe"] for r in out] == [i % 4 == 0 for i in range(n_frames)]
    assert all(n >= 100 for n in n3), n3                                  # stereo matches became map points
    for i in range(1, n_frames):
        assert out[i]["status"].all(), (i, out[i]["status"])              # compute_pose! accepted in every stream
        assert (out[i]["counts"] >= 150).all(), (i, out[i]["counts"])     # the lists survive tracking and both outlier filters
        for s in range(S):
            o = offs[s][i] - offs[s][0]                                   # image offset (y, x) since frame 0
            want = np.array([o[1] * Z / cam[0], o[0] * Z / cam[1], 0.0])
            T = out[i]["poses"][s]
            assert np.abs(T[:3, :3] - np.eye(3)).max() < 2e-3, (i, s)
            assert np.abs(T[:3, 3] - want).max() < 0.03 * max(1.0, np.abs(want).max()) + 0.02, (i, s, T[:3, 3], want)
    # the five-point filter ran (enough parallax against the key-frame) at least once per stream
    assert all(any(r["status_5pt"][s] for r in out[1:]) for s in range(S))
