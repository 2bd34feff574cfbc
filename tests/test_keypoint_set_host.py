"""CPU: the host-side restatements that accompany the device-resident pose seams (slam.jl_amd/keypoint_set.py): the tuple
generator the kernels share with the tests, the gather arithmetic against the oracle's camera model, the pose composition."""
import numpy as np


def test_pose_samples_are_distinct_deterministic_and_in_range(slam_host):
    a = slam_host.pose_samples(77, 3, 41, 200)
    b = slam_host.pose_samples(77, 3, 41, 200)
    assert np.array_equal(a, b) and a.dtype == np.int32 and a.shape == (200, 3)
    assert a.min() >= 0 and a.max() < 41
    assert all(len(set(r)) == 3 for r in a.tolist())
    assert not np.array_equal(a, slam_host.pose_samples(78, 3, 41, 200)) and not np.array_equal(a, slam_host.pose_samples(77, 4, 41, 200))
    assert (slam_host.pose_samples(1, 0, 4, 8) == -1).all()                    # fewer than five 3-D keypoints: no P3P (front_end.jl:133)
    f = slam_host.pose_samples5(5, 1, 9, 64)
    assert f.shape == (64, 5) and f.min() >= 0 and f.max() < 9 and all(len(set(r)) == 5 for r in f.tolist())
    # the first draw of a stream is the plain generator value: splitmix64(seed ^ stream << 48 ^ it << 16 ^ attempt) mod n
    from slam_jl_amd.keypoint_set import _splitmix64
    assert a[0, 0] == _splitmix64(77 ^ (3 << 48)) % 41
    assert _splitmix64(0) == 0xE220A8397B1DCDAF                            # reference value of the published generator


def test_pose_inputs_follow_the_camera_model(slam_host, orc, syn):
    cam = syn.KITTI_CAM
    dist = (-0.28, 0.07, 2e-4, -1e-4)
    rng = np.random.default_rng(2)
    yx = np.stack([rng.uniform(1, 370, 50), rng.uniform(1, 1226, 50)], axis=1)
    xyz = rng.normal(0, 5, (50, 3))
    pts, px, pdn = slam_host.pose_inputs(cam, dist, yx, xyz)
    assert np.array_equal(pts, xyz)
    for i in range(50):
        u = orc.undistort_point(cam, dist, yx[i])                          # camera.jl:98-125, (y, x)
        assert np.allclose(px[i], u[::-1], rtol=0, atol=1e-12)             # P3P takes (x, y): front_end.jl:151
        b = np.array([(u[1] - cam[2]) / cam[0], (u[0] - cam[3]) / cam[1], 1.0])     # backproject, camera.jl:138-140
        assert np.allclose(pdn[i], b / np.linalg.norm(b), rtol=0, atol=1e-15)
    p1, p2, d1, d2 = slam_host.pose_5pt_inputs(cam, dist, yx, yx + 3.0)
    assert np.allclose(p2, px) and np.allclose(d2[:, 0], (px[:, 0] - cam[2]) / cam[0]) and p1.shape == p2.shape == d1.shape == (50, 2)


def test_pose_5pt_compose_scales_and_chains(slam_host):
    th = 0.1
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    Rt = np.concatenate([R, np.array([[0.6], [0.0], [0.8]])], axis=1)      # |t| = 1
    prev_cw = np.eye(4); prev_cw[:3, 3] = [1.0, 2.0, 3.0]
    cur_wc = np.eye(4); cur_wc[:3, 3] = [-1.0, -2.0, -0.5]                 # motion model: 2.5 m from the key-frame (front_end.jl:322-324)
    T = slam_host.pose_5pt_compose(Rt, prev_cw, cur_wc)
    rel = T @ np.linalg.inv(prev_cw)
    assert np.allclose(rel[:3, :3], R) and np.allclose(rel[:3, 3], 2.5 * np.array([0.6, 0.0, 0.8]))
