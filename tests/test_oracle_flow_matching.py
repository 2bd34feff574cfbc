"""CPU: the oracle's restatement of optical_flow_matching! (map_manager.jl:451-564, maybe_stereo_update! :579-590)
against its own building block (orc.fb_tracking) on constructed cases: which keypoints go through which pass, which
are skipped / removed, and what the stereo gate writes."""
import numpy as np


def _setup(orc, texture, H=120, W=160, disparity=12.4):
    L, R, flows = texture(H, W, disparity=disparity)
    kp = orc.detect(L[0], np.zeros((0, 2)), max_points=120).astype(float)
    return L, R, flows, kp


def test_temporal_protocol(orc, texture):
    H, W = 120, 160
    L, R, flows, kp = _setup(orc, texture)
    a, b = orc.pyr_build(L[0], 3, 1.0, 1), orc.pyr_build(L[1], 3, 1.0, 1)
    n = len(kp)
    is3 = np.arange(n) % 2 == 0
    proj = kp + np.array(flows[1])
    proj[4] = (H + 3.0, 20.0)                        # 3-D keypoint projected outside the image: skipped entirely (:501-506)
    proj[6] += 40.0                                  # bad prior: the 3-D attempt fails, re-tracked with the 2-D set (:533-538)
    r = orc.optical_flow_matching(a, b, kp, is3, proj, (H, W))
    assert not r["updated"][4] and not r["removed"][4] and np.array_equal(r["new_pixels"][4], kp[4])
    assert not (r["updated"] & r["removed"]).any()
    inside = (proj[:, 0] >= 1) & (proj[:, 0] <= H) & (proj[:, 1] >= 1) & (proj[:, 1] <= W)
    assert np.array_equal(r["updated"] | r["removed"], ~(is3 & ~inside))
    # 2-D keypoints: exactly fb_tracking! without prior on 3 levels
    i2 = np.where(~is3)[0]
    nk, st = orc.fb_tracking(a, b, kp[i2], pyramid_levels=3, window=9, max_distance=1.0)
    assert np.array_equal(r["updated"][i2], st) and np.array_equal(r["new_pixels"][i2][st], nk[st])
    # 3-D keypoints with a good prior: one level, prior (proj - px) / 2
    i3 = np.array([j for j in np.where(is3 & inside)[0] if j not in (4, 6)])
    nk3, st3 = orc.fb_tracking(a, b, kp[i3], disp0=0.5 * (proj[i3] - kp[i3]), pyramid_levels=1, window=9, max_distance=1.0)
    assert st3.mean() > 0.6
    assert np.array_equal(r["new_pixels"][i3][st3], nk3[st3]) and r["updated"][i3][st3].all()
    # the bad prior fell back to the 2-D pass and equals a prior-free track
    nk6, st6 = orc.fb_tracking(a, b, kp[[6]], pyramid_levels=3, window=9, max_distance=1.0)
    assert r["updated"][6] == st6[0] and (not st6[0] or np.array_equal(r["new_pixels"][6], nk6[0]))
    assert np.abs(np.median((r["new_pixels"] - kp)[r["updated"]], 0) - np.array(flows[1])).max() < 0.05


def test_stereo_protocol_and_epipolar_gate(orc, texture, syn):
    H, W = 120, 160
    L, R, flows, kp = _setup(orc, texture, disparity=6.3)
    a, b = orc.pyr_build(L[0], 3, 1.0, 1), orc.pyr_build(R[0], 3, 1.0, 1)
    n = len(kp)
    is3 = np.arange(n) % 3 == 0
    proj = kp + np.array([0.0, -6.3])
    proj[3] = (10.0, -5.0)                           # outside the right image: observation removed (:491-498)
    und = kp.copy()
    und[5, 0] += 7.0                                 # pretend the left keypoint's undistorted row is 7 px away: gate rejects
    cam = syn.KITTI_CAM
    r = orc.optical_flow_matching(a, b, kp, is3, proj, (H, W), stereo=True, undistorted_left=und, right_cam=cam)
    assert r["removed"][3] and not r["updated"][3]
    inside = (proj[:, 0] >= 1) & (proj[:, 0] <= H) & (proj[:, 1] >= 1) & (proj[:, 1] <= W)
    assert np.array_equal(r["removed"], is3 & ~inside)   # stereo matching never removes on a failed track (:553-555)
    assert not r["updated"][5]
    up = r["updated"]
    assert up.mean() > 0.5
    assert np.array_equal(r["new_pixels"][up][:, 0], kp[up][:, 0])          # row of the left keypoint kept (:587)
    assert abs(np.median((r["new_pixels"] - kp)[up][:, 1]) + 6.3) < 0.1


def test_undistort_point_is_identity_without_distortion(orc, syn):
    cam = syn.KITTI_CAM
    rng = np.random.default_rng(0)
    for p in rng.uniform(1, 370, (50, 2)):
        assert np.abs(orc.undistort_point(cam, (0, 0, 0, 0), p) - p).max() < 1e-10
    # with distortion: first-order check against the closed form of the radial term
    q = orc.undistort_point(cam, (0.1, 0.0, 0.0, 0.0), np.array([100.0, 200.0]))
    fx, fy, cx, cy = cam
    ny, nx = (100.0 - cy) / fy, (200.0 - cx) / fx
    rd = 1 + 0.1 * (ny * ny + nx * nx)
    assert np.allclose(q, [rd * ny * fy + cy, rd * nx * fx + cx], atol=1e-12)
