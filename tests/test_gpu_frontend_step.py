"""GPU: slam_frontend_step -- one C call per frame for a live stream (reference: src/front_end.jl:58-113, :454-470; key-frames
src/map_manager.jl:98-113, src/mapper.jl:51-66, :142-183).  The call enqueues exactly the seams a host would call one by one, so after every
frame its keypoint list must equal, to the bit, the list of a KeypointSet driven through those seams by hand on single pyramids
(slam_pyr_update_u8 + slam_kpset_*), which tests/test_gpu_kpset.py holds to the oracle; the look-ahead mode (build of frame t + 1 during
the matching of frame t) must produce the same lists one call later."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H, W = 185, 300
N_FRAMES, KF = 11, 5
DISP = 6.4


def _stream(syn):
    L, R, flows = syn.stereo_stream((H, W), N_FRAMES, seed=7, step=(1.1, -1.7), disparity=DISP)
    u8 = lambda im: np.ascontiguousarray(np.round(np.clip(im, 0, 1) * 255).astype(np.uint8).T)      # column-major bytes
    return [u8(x) for x in L], [u8(x) for x in R], np.array(flows)


def _by_hand(slam, syn, Lf, Rf, flows, fast):
    """the same frames through the single-image seams + a one-stream KeypointSet, one call at a time"""
    params = slam.Params(stereo=True, max_nb_keypoints=300)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    ex = slam.Extractor.from_params(params, cam)
    ctx = slam.default_context(0)
    cap = ex.max_points + ex.grid_resolution[0] * ex.grid_resolution[1] + 64
    ks = slam.KeypointSet(1, cap)
    pyr = [slam.LKPyramid(shape=(H, W), levels=params.pyramid_levels) for _ in range(3)]
    rp = slam.LKPyramid(shape=(H, W), levels=params.pyramid_levels)
    camt = tuple(syn.KITTI_CAM)
    T21 = np.eye(4); T21[0, 3] = -0.54
    lists, n = [], 0
    mode = 3 if fast else 1

    class B:                                               # (KeypointSet's calls take batch-like objects: .pyramids[0])
        def __init__(self, p): self.pyramids = [p]
    for t in range(N_FRAMES):
        cur, prev = pyr[t % 3], pyr[(t + 2) % 3]
        ctx.check(ctx.lib.slam_pyr_update_u8(ctx.h, cur.h, Lf[t].ctypes.data_as(slam._lib.u8p), mode, 1.0))
        if t > 0 and n > 0:
            ks.flow_match(B(prev), B(cur), params, slam.stream_params(1, cam=camt, shift_yx=flows[t] - flows[t - 1]), prior=2, n_bound=n)
        if t % KF == 0:
            ks.detect(ex, B(cur)); ks.keyframe()
            ctx.check(ctx.lib.slam_pyr_update_u8(ctx.h, rp.h, Rf[t].ctypes.data_as(slam._lib.u8p), mode | 16, 1.0))
            ks.stereo_match(B(cur), B(rp), params, slam.stream_params(1, cam=camt, shift_yx=(0.0, -DISP)), prior=2)
            ks.triangulate(camt, camt, T21, np.eye(4), max_error=3.0)
        n = int(ks.counts()[0])
        lists.append(ks.download(0))
    ks.close()
    return params, ex, camt, T21, lists


@pytest.mark.parametrize("fast", [False, True])
def test_step_equals_the_seams_called_one_by_one(slam, syn, fast):
    Lf, Rf, flows = _stream(syn)
    params, ex, camt, T21, want = _by_hand(slam, syn, Lf, Rf, flows, fast)
    tri = slam.FrontEnd.tri_params(camt, camt, T21, np.eye(4))
    sps = slam.stream_params(1, cam=camt, shift_yx=(0.0, -DISP))
    for lookahead in (False, True):
        fe = slam.FrontEnd((H, W), params, ex, fast=fast, lookahead=lookahead)
        got = {}
        def args(t):                                       # the arguments of the frame a call PROCESSES
            return dict(params=slam.stream_params(1, cam=camt, shift_yx=flows[t] - flows[t - 1]) if t > 0 else slam.stream_params(1, cam=camt), prior=2,
                        stereo_params=sps, stereo_prior=2, tri=tri)
        for t in range(N_FRAMES):
            due = t - 1 if lookahead else t
            fr, cnt = fe.step(Lf[t], Rf[t] if t % KF == 0 else None, **args(max(due, 0)))
            assert fr == due if due >= 0 else fr == -1
            if fr >= 0:
                got[fr] = (cnt, fe.keypoints())
        if lookahead:
            fr, cnt = fe.flush(**args(N_FRAMES - 1)); assert fr == N_FRAMES - 1
            got[fr] = (cnt, fe.keypoints())
            assert fe.flush(**args(N_FRAMES - 1))[0] == -1          # nothing left in flight
        for t in range(N_FRAMES):
            cnt, g = got[t]; w = want[t]
            assert cnt == len(w["yx"]) and cnt > 50, (lookahead, t, cnt)
            for k in ("yx", "is_3d", "ids", "has_stereo"):
                assert np.array_equal(g[k], w[k]), (lookahead, t, k)
            assert np.array_equal(g["xyz"][g["is_3d"]], w["xyz"][w["is_3d"]]) and np.array_equal(g["stereo_yx"][g["has_stereo"]], w["stereo_yx"][w["has_stereo"]])
        assert got[N_FRAMES - 1][1]["is_3d"].sum() > 20      # stereo matches were triangulated
        fe.close()


def test_bad_configuration_and_null_frame_are_argument_errors(slam, syn):
    params = slam.Params(stereo=True, max_nb_keypoints=300)
    ex = slam.Extractor.from_params(params, slam.Camera(*syn.KITTI_CAM, height=H, width=W))
    with pytest.raises(slam.SlamHipError, match="bad configuration"):
        slam.FrontEnd((H, W), params, ex, cap=10)
    fe = slam.FrontEnd((H, W), params, ex)
    assert fe.lib.slam_frontend_step(fe.h, None, None, None, 0, None, 0, None, None, None, None) == -1
    fe.close()
