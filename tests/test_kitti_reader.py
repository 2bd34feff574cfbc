"""CPU: the KITTI odometry reader (slam.jl_amd/kitti.py) against the reference's semantics (example/kitty/kitty.jl)
on a synthetic KITTI-shaped directory, and the zlib PNG path against PIL."""
import io
import os

import numpy as np
import pytest


def test_parse_and_pose_round_trip(slam_host, tmp_path):
    from slam_jl_amd import kitti
    T = np.eye(4); T[:3, :4] = np.arange(12).reshape(3, 4) * 0.25 - 1
    kitti.write_poses(tmp_path / "p.txt", [T, np.eye(4)])
    back = kitti.read_poses(tmp_path / "p.txt")
    assert len(back) == 2 and np.allclose(back[0], T, atol=1e-12) and np.array_equal(back[1], np.eye(4))
    with pytest.raises(ValueError):
        kitti.parse_matrix("1 2 3")


def test_png_codec_matches_pil(slam_host):
    from slam_jl_amd import kitti
    from PIL import Image
    rng = np.random.default_rng(0)
    img = (rng.random((37, 53)) * 255).astype(np.uint8)
    img[5:20, 10:40] = np.linspace(0, 255, 30).astype(np.uint8)              # smooth ramp: PIL picks non-trivial filters
    assert np.array_equal(kitti.decode_png_gray8(kitti.encode_png_gray8(img)), img)
    buf = io.BytesIO(); Image.fromarray(img, mode="L").save(buf, format="PNG", optimize=True)
    assert np.array_equal(kitti.decode_png_gray8(buf.getvalue()), img)                      # PIL-written, all filter types
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(kitti.encode_png_gray8(img)))), img)   # ours, PIL-read
    with pytest.raises(ValueError):
        kitti.decode_png_gray8(b"not a png at all")
    rgb = io.BytesIO(); Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(rgb, format="PNG")
    with pytest.raises(ValueError):
        kitti.decode_png_gray8(rgb.getvalue())


def test_dataset_matches_reference_semantics(slam_host, syn, tmp_path):
    from slam_jl_amd import kitti
    L, R, flows = syn.stereo_stream((60, 90), 3, seed=5, disparity=4.0)
    cam = syn.KITTI_CAM
    poses = [np.eye(4) for _ in range(3)]
    for i, P in enumerate(poses):
        P[2, 3] = 0.8 * i
    kitti.write_synthetic_sequence(str(tmp_path), "05", L, R, cam, 0.54, poses)
    d = slam_host.KittyDataset(str(tmp_path), "05", stereo=True)
    assert len(d) == 3 and len(d.timestamps) == 3 and d.timestamps[1] == pytest.approx(0.1)
    assert np.allclose(d.intrinsics, cam) and d.K.shape == (4, 4) and d.K[3, 3] == 1.0
    # Ti0 = inv(K) * P1: a pure -baseline shift along x, everything else exactly identity (kitty.jl:58-62)
    want = np.eye(4); want[0, 3] = -0.54
    assert np.allclose(d.Ti0, want, atol=1e-12) and d.baseline == pytest.approx(0.54)
    a, b = d[1]
    assert a.dtype == np.uint8 and a.shape == (60, 90)
    assert np.array_equal(a, np.round(L[1] * 255).astype(np.uint8)) and np.array_equal(b, np.round(R[1] * 255).astype(np.uint8))
    mono = slam_host.KittyDataset(str(tmp_path), "05", stereo=False)
    m0, m1 = mono[2]
    assert m0 is m1
    pos, dirs = d.get_camera_poses()
    assert np.allclose(pos[:, 2], [0, 0.8, 1.6]) and np.allclose(dirs, [[0, 0, 1]] * 3)
    with pytest.raises(IndexError):
        d[3]
    os.remove(os.path.join(str(tmp_path), "poses", "05.txt"))                # test sequences ship no ground truth
    assert len(slam_host.KittyDataset(str(tmp_path), "05")) == 3
