"""GPU: the device-resident keypoint set (slam_kpset_*) against the host protocol it replaces -- the same steps through the
batch seams with numpy list surgery in between (slam_flow_match_batch_kept, numpy cull, slam_detect_batch + merge,
slam_flow_match_batch for the stereo pair, slam_triangulate) -- element for element, and against the CPU oracle's
optical_flow_matching! restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(slam, texture, S, H, W, levels=3):
    import torch
    streams = [texture(H, W, seed=30 + s, step=(1.0 + 0.2 * s, -1.4), disparity=6.3) for s in range(S)]
    a = slam.PyramidBatch((H, W), levels=levels, S=S); b = slam.PyramidBatch((H, W), levels=levels, S=S); r = slam.PyramidBatch((H, W), levels=levels, S=S)
    dev = lambda k, f: [torch.from_numpy(np.ascontiguousarray(st[k][f].T)).cuda() for st in streams]
    d0, d1, dr = dev(0, 0), dev(0, 1), dev(1, 1)
    torch.cuda.synchronize()
    a.update_([d.data_ptr() for d in d0]); b.update_([d.data_ptr() for d in d1]); r.update_([d.data_ptr() for d in dr])
    return streams, a, b, r, (d0, d1, dr)


def test_keyframe_step_equals_host_protocol(slam, orc, syn, texture):
    import torch
    S, H, W = 3, 120, 160
    streams, a, b, r, keep = _setup(slam, texture, S, H, W)
    params = slam.Params(stereo=True, max_nb_keypoints=150)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    e = slam.Extractor.from_params(params, cam)
    ncell = e.grid_resolution[0] * e.grid_resolution[1]
    cap = params.max_nb_keypoints + ncell + 8
    rng = np.random.default_rng(5)
    # initial lists: detected keypoints of frame 0, some 3-D, some at hopeless places (borders), ragged counts
    kps, is3, sid = [], [], []
    for s in range(S):
        k = orc.detect(streams[s][0][0], np.zeros((0, 2)), max_points=60 + 20 * s).astype(float)
        k = np.concatenate([k, np.array([[1.0, 1.0], [H, W], [2.5, W - 1.5]])])
        kps.append(k); is3.append(rng.random(len(k)) < 0.5); sid.append(np.full(len(k), s, np.int32))
    ks = slam.KeypointSet(S, cap)
    for s in range(S):
        ks.upload(s, kps[s], is3[s])
    shift = np.array([streams[s][2][1] for s in range(S)])                       # per-stream prior shift (the true flow)
    sp = slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=shift)
    # ---- temporal match ----
    ks.flow_match(a, b, params, sp, prior=2)
    P = np.concatenate(kps); T = np.concatenate(is3); I = np.concatenate(sid)
    proj = P + shift[I]
    inside = (proj[:, 0] >= 1) & (proj[:, 0] <= H) & (proj[:, 1] >= 1) & (proj[:, 1] <= W)
    skip = T & ~inside                                                           # map_manager.jl:501-506: left as they are
    hk, h3, hs, hsrc = slam.optical_flow_matching_batch_kept(a, b, I[~skip], P[~skip], T[~skip], proj[~skip], params)
    # host list surgery: survivors + skipped ones, in input order
    pos = np.full((len(P), 2), np.nan); alive = np.zeros(len(P), bool)
    idx_ns = np.flatnonzero(~skip)
    pos[idx_ns[hsrc]] = hk; alive[idx_ns[hsrc]] = True
    pos[skip] = P[skip]; alive[skip] = True
    cnt = ks.counts()
    for s in range(S):
        m = (I == s) & alive
        got = ks.download(s)
        assert cnt[s] == m.sum() == len(got["yx"]), s
        assert np.array_equal(got["yx"], pos[m]) and np.array_equal(got["is_3d"], T[m]), s
        # and against the CPU oracle's optical_flow_matching! restatement
        ra, rb = orc.pyr_build(streams[s][0][0], 3, 1.0, 1), orc.pyr_build(streams[s][0][1], 3, 1.0, 1)
        ref = orc.optical_flow_matching(ra, rb, kps[s], is3[s], kps[s] + shift[s], (H, W), sum_order=1)
        keep_ref = ~ref["removed"]
        assert np.array_equal(keep_ref, alive[I == s]), s
        assert np.abs(got["yx"] - ref["new_pixels"][keep_ref]).max() <= 1e-9, s
    P, T, I = pos[alive], T[alive], I[alive]
    # ---- cull (flags in HBM) ----
    flags = np.zeros((S, cap), np.uint8)
    for s in range(S):
        n = int(cnt[s]); flags[s, :n] = rng.random(n) < 0.2
    fdev = torch.from_numpy(flags).cuda(); torch.cuda.synchronize()
    ks.remove(fdev.data_ptr())
    keepm = np.concatenate([flags[s, :int(cnt[s])] == 0 for s in range(S)])
    P, T, I = P[keepm], T[keepm], I[keepm]
    # ---- detect + merge ----
    ks.detect(e, b)
    fresh, fsid = slam.detect_batch(e, b, P, I)
    cnt2 = ks.counts()
    lists = []
    for s in range(S):
        cur = P[I == s]; new = fresh[fsid == s].astype(float)
        lists.append((np.concatenate([cur, new]), np.concatenate([T[I == s], np.zeros(len(new), bool)])))
        got = ks.download(s)
        assert cnt2[s] == len(lists[s][0]), s
        assert np.array_equal(got["yx"], lists[s][0]) and np.array_equal(got["is_3d"], lists[s][1]), s
        assert len(np.unique(got["ids"])) == len(got["ids"])
    assert sum(len(l[0]) for l in lists) > len(P)                                # something was detected
    # ---- stereo match (left = b, right = r) + epipolar gate ----
    sps = slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=np.tile([0.0, -6.3], (S, 1)))
    ks.stereo_match(b, r, params, sps, prior=2)
    for s in range(S):
        kp, t3 = lists[s]
        res = slam.optical_flow_matching_frame(b.pyramids[s], r.pyramids[s], kp, t3, kp + np.array([0.0, -6.3]), params, (H, W), stereo=True,
                                               undistorted_left=kp, right_cam=syn.KITTI_CAM)
        got = ks.download(s)
        keep_s = ~res["removed"]
        assert np.array_equal(got["yx"], kp[keep_s]), s                          # positions untouched, out-of-image 3-D observations removed
        assert np.array_equal(got["has_stereo"], res["updated"][keep_s]), s
        up = got["has_stereo"]
        assert np.array_equal(got["stereo_yx"][up], res["new_pixels"][keep_s][up]), s
        assert up.mean() > 0.3
        lists[s] = (kp[keep_s], t3[keep_s], got["stereo_yx"], up)
    # ---- triangulation of the 2-D keypoints with a stereo match ----
    T21 = np.eye(4); T21[0, 3] = -0.54                                           # right camera 0.54 m to the right of the left one
    Twc = np.eye(4); Twc[:3, 3] = [1.0, 2.0, 3.0]
    ks.triangulate(syn.KITTI_CAM, syn.KITTI_CAM, T21, Twc, max_error=3.0)
    for s in range(S):
        kp, t3, syx, up = lists[s]
        got = ks.download(s)
        cand = up & ~t3
        xyz, ok = slam.triangulate(syn.KITTI_CAM, syn.KITTI_CAM, T21, kp[cand], syx[cand], 3.0)
        exp3 = t3.copy(); exp3[np.flatnonzero(cand)[ok]] = True
        exps = up.copy(); exps[np.flatnonzero(cand)[~ok]] = False
        assert np.array_equal(got["is_3d"], exp3) and np.array_equal(got["has_stereo"], exps), s
        world = xyz[ok] + Twc[:3, 3]
        assert np.abs(got["xyz"][np.flatnonzero(cand)[ok]] - world).max() <= 1e-9 * max(1.0, np.abs(world).max()), s
        assert ok.any()


def test_pose_prior_and_capacity(slam, syn, texture):
    """prior = 1: the projection of the map point through Tcw and the lens model; full lists are not extended by detect."""
    S, H, W = 2, 120, 160
    streams, a, b, r, keep = _setup(slam, texture, S, H, W)
    params = slam.Params(stereo=True, max_nb_keypoints=40)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    e = slam.Extractor.from_params(params, cam)
    cap = 40 + e.grid_resolution[0] * e.grid_resolution[1]
    fx, fy, cx, cy = syn.KITTI_CAM
    rng = np.random.default_rng(1)
    ks = slam.KeypointSet(S, cap)
    kp = np.stack([rng.uniform(25, H - 25, 45), rng.uniform(25, W - 25, 45)], 1)
    z = rng.uniform(5, 30, 45)
    flow = np.array(streams[0][2][1])
    tgt = kp + flow                                                              # where the map point should project in the new frame
    xyz = np.stack([(tgt[:, 1] - cx) / fx * z, (tgt[:, 0] - cy) / fy * z, z], 1)  # camera = world (Tcw = I)
    for s in range(S):
        ks.upload(s, kp, np.ones(45, bool), xyz=xyz)
    sp = slam.stream_params(S, Tcw=np.eye(4), cam=syn.KITTI_CAM)
    ks.flow_match(a, b, params, sp, prior=1)
    got = ks.download(0)
    new, st = slam.optical_flow_matching(a.pyramids[0], b.pyramids[0], kp, np.ones(45, bool), tgt, params)
    assert st.sum() == len(got["yx"]) and np.abs(got["yx"] - new[st]).max() <= 1e-9    # the device projects xyz itself: the prior agrees to rounding
    # stream 1 is at / above max_points after re-upload: detect must not append to it
    ks.upload(0, kp[:10], np.zeros(10, bool)); ks.upload(1, kp[:41], np.zeros(41, bool))
    ks.detect(e, b)
    c = ks.counts()
    assert c[1] == 41 and c[0] > 10
    ref = slam.detect(e, b.pyramids[0], kp[:10])
    assert np.array_equal(ks.download(0)["yx"][10:], ref.astype(float))


def test_compute_pose_on_the_set_equals_the_host_seams(slam, syn):
    """slam_kpset_compute_pose (front_end.jl:132-219 on the device-resident lists) against the host route it replaces: download
    the lists, build the P3P inputs with numpy, the SAME triples (pose_samples), slam_p3p_ransac_batch, numpy inlier surgery,
    slam_pnp_ba_batch, numpy removal -- poses, status words and the surviving lists."""
    S, cap = 4, 700
    ks = slam.KeypointSet(S, cap)
    cam = syn.KITTI_CAM
    rng = np.random.default_rng(11)
    host = []
    for s in range(S):
        n3 = [400, 3, 250, 60][s]                               # stream 1: fewer than five 3-D keypoints -> status 0, nothing removed
        sc = syn.p3p_scene(n=n3, seed=20 + s, noise_px=0.3, outlier_frac=[0.25, 0.0, 0.1, 0.9][s], iters=4)   # stream 3: hardly any inliers
        n2 = 50 + 10 * s                                         # 2-D keypoints interleaved with the 3-D ones
        n = n3 + n2
        order = rng.permutation(n)
        yx = np.zeros((n, 2)); is3 = np.zeros(n, bool); xyz = np.zeros((n, 3))
        yx[order[:n3]] = sc["px_xy"][:, ::-1]; is3[order[:n3]] = True; xyz[order[:n3]] = sc["pts3d"]
        yx[order[n3:]] = rng.uniform(5, 300, (n2, 2))
        ks.upload(s, yx, is3, xyz)
        host.append((yx, is3, xyz, sc))
    dist = (0.0, 0.0, 0.0, 0.0)
    sp = slam.stream_params(S, cam=cam, dist=dist)
    before = [ks.download(s) for s in range(S)]
    iters, seed, thr = 128, 77, 3.0
    poses, status, ninl, counts = ks.compute_pose(sp, threshold=thr, iters=iters, seed=seed)
    after = [ks.download(s) for s in range(S)]
    # ---- the host route ----
    K = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1.0]])
    P, X, B, SM, idx3 = [], [], [], [], []
    for s in range(S):
        d = before[s]
        m = d["is_3d"].astype(bool)
        pts, px, pdn = slam.pose_inputs(cam, dist, d["yx"][m], d["xyz"][m])
        P.append(px); X.append(pts); B.append(pdn); idx3.append(np.flatnonzero(m))
        SM.append(slam.pose_samples(seed, s, int(m.sum()), iters))
    r3 = slam.p3p_ransac_batch(X, P, B, K, threshold=thr, samples=SM)
    ok = [r is not None and len(X[s]) >= 5 and r[0] >= 5 for s, r in enumerate(r3)]
    T0, bp, bx = [], [], []
    for s in range(S):
        if ok[s]:
            inl = r3[s][1][1]
            T = np.eye(4); T[:3] = r3[s][1][3]
            T0.append(T); bp.append(P[s][inl][:, ::-1]); bx.append(X[s][inl])
        else:
            T0.append(np.eye(4)); bp.append(np.zeros((0, 2))); bx.append(np.zeros((0, 3)))
    rb = slam.pnp_bundle_adjustment_batch(cam, T0, bp, bx, repr_eps=thr)
    assert ok == [True, False, True, False] or ok == [True, False, True, True]
    for s in range(S):
        keep = np.ones(len(before[s]["yx"]), bool)
        accept = False
        if ok[s]:
            inl = r3[s][1][1]
            keep[idx3[s][~inl]] = False                                          # P3P outliers leave the list
            newT, e0, e1, outl, no = rb[s]
            accept = not (int(inl.sum()) - no < 5 or e1 > e0) and not np.array_equal(newT, np.eye(4))
            if accept:
                keep[idx3[s][inl][outl]] = False
                assert np.allclose(poses[s], newT, rtol=0, atol=1e-9), (s, np.abs(poses[s] - newT).max())
                assert ninl[s] == int(inl.sum())
        assert status[s] == (1 if accept else 0), s
        if not accept:
            assert np.array_equal(poses[s], np.eye(4))
        assert counts[s] == keep.sum() == len(after[s]["yx"]), (s, counts[s], keep.sum())
        assert np.array_equal(after[s]["yx"], before[s]["yx"][keep]) and np.array_equal(after[s]["ids"], before[s]["ids"][keep]), s
        assert np.array_equal(after[s]["is_3d"], before[s]["is_3d"][keep])
    # the accepted poses are the scene's pose
    for s in (0, 2):
        assert status[s] == 1
        Rt = host[s][3]["Rt_gt"]
        assert np.abs(poses[s][:3] - Rt).max() < 5e-3, np.abs(poses[s][:3] - Rt).max()


def test_compute_pose_5pt_on_the_set_equals_the_host_seam(slam, syn):
    """slam_kpset_compute_pose_5pt (front_end.jl:242-332 on the device-resident lists + the key-frame observation every keypoint
    carries) against the host route: download, numpy gather of the pairs the key-frame observes, the SAME 5-tuples
    (pose_samples5), slam_five_point_ransac_batch, numpy removal -- [R | t], status words, parallax and the surviving lists."""
    S, cap = 4, 600
    ks = slam.KeypointSet(S, cap)
    cam = syn.KITTI_CAM
    dist = (0.0, 0.0, 0.0, 0.0)
    rng = np.random.default_rng(3)
    for s in range(S):
        n5 = [300, 6, 200, 120][s]                              # stream 1: fewer than 8 pairs -> status 0
        fs = syn.five_point_scene(n=n5, seed=60 + s, noise_px=0.3, outlier_frac=[0.2, 0.0, 0.1, 0.3][s], iters=4)
        extra = 15 + 5 * s                                       # keypoints the key-frame does not observe, interleaved
        n = n5 + extra
        order = rng.permutation(n)
        yx = rng.uniform(5, 300, (n, 2)); kyx = np.zeros((n, 2)); hk = np.zeros(n, bool)
        yx[order[:n5]] = fs["px2"][:, ::-1]; kyx[order[:n5]] = fs["px1"][:, ::-1]; hk[order[:n5]] = True
        if s == 3:
            kyx[hk] = yx[hk] + 0.25                              # no parallax: below min_parallax -> status 0
        ks.upload(s, yx, np.zeros(n, bool))
        ks.upload_keyframe(s, kyx, hk)
    Rc = np.eye(4)
    sp = slam.stream_params(S, Tcw=Rc, cam=cam, dist=dist)
    before = [ks.download(s) for s in range(S)]
    kf = [ks.download_keyframe(s) for s in range(S)]
    iters, seed, thr = 96, 5, 3.0
    Rt, status, ninl, par, counts = ks.compute_pose_5pt(sp, min_parallax=5.0, max_repr_error=thr, iters=iters, seed=seed)
    after = [ks.download(s) for s in range(S)]
    K = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1.0]])
    A1, A2, D1, D2, SM, idx, avg = [], [], [], [], [], [], []
    for s in range(S):
        m = kf[s][1]
        p1, p2, d1, d2 = slam.pose_5pt_inputs(cam, dist, before[s]["yx"][m], kf[s][0][m])
        A1.append(p1); A2.append(p2); D1.append(d1); D2.append(d2); idx.append(np.flatnonzero(m))
        # identity compensation, no distortion: project(position) is the undistorted pixel itself (up to rounding)
        avg.append(np.linalg.norm(p2 - p1, axis=1).mean() if m.sum() else 0.0)
        run = len(before[s]["yx"]) >= 8 and m.sum() >= 8 and not (avg[-1] < 5.0)
        SM.append(slam.pose_samples5(seed, s, int(m.sum()), iters) if run else np.full((iters, 5), -1, np.int32))
    r5 = slam.five_point_ransac_batch(A1, A2, D1, D2, K, K, max_repr_error=thr, samples=SM)
    expect_ok = []
    for s in range(S):
        keep = np.ones(len(before[s]["yx"]), bool)
        n_in = r5[s][0]
        ok = (SM[s][0, 0] >= 0) and n_in >= 5
        expect_ok.append(bool(ok))
        assert status[s] == (1 if ok else 0), (s, status[s], n_in)
        assert abs(par[s] - avg[s]) <= 1e-9 * max(1.0, avg[s]), (s, par[s], avg[s])
        if ok:
            inl = r5[s][1][2]
            if n_in != len(inl):
                keep[idx[s][~inl]] = False
            assert ninl[s] == n_in
            assert np.array_equal(Rt[s], r5[s][1][1]), (s, np.abs(Rt[s] - r5[s][1][1]).max())
        assert counts[s] == keep.sum() == len(after[s]["yx"]), (s, counts[s], keep.sum())
        assert np.array_equal(after[s]["ids"], before[s]["ids"][keep]) and np.array_equal(after[s]["yx"], before[s]["yx"][keep]), s
        k2, h2 = ks.download_keyframe(s)
        assert np.array_equal(h2, kf[s][1][keep]) and np.array_equal(k2[h2], kf[s][0][keep][h2]), s      # the observation travels with the keypoint
    assert expect_ok == [True, False, True, False]
    # key-frame snapshot: every live keypoint gets its current position
    ks.keyframe()
    for s in range(S):
        k2, h2 = ks.download_keyframe(s)
        assert h2.all() and np.array_equal(k2, after[s]["yx"])


def test_target_only_right_pyramids_give_the_same_stereo_matches(slam, syn, texture):
    """SLAM_PYR_TARGET_ONLY: the right batch of a stereo match built with layers only above level 0 -- the matches, the removals and
    every plane a target is ever read from are identical to the full build; using such a batch as the source is refused."""
    S, H, W = 3, 160, 240
    streams, a, b, r, keep = _setup(slam, texture, S, H, W)
    import torch
    r2 = slam.PyramidBatch((H, W), levels=3, S=S)
    r2.update_([d.data_ptr() for d in keep[2]], target_only=True)
    torch.cuda.synchronize()
    for s in range(S):
        for lvl in range(4):                                                   # 0-based: every layer
            assert np.array_equal(r2.pyramids[s].plane("layers", lvl), r.pyramids[s].plane("layers", lvl)), (s, lvl)
        for name in ("Iy", "Ix", "Iyy", "Ixx", "Iyx"):                             # the finest level's gradient / integral planes
            assert np.array_equal(r2.pyramids[s].plane(name, 0), r.pyramids[s].plane(name, 0)), (s, name)
    params = slam.Params(stereo=True, max_nb_keypoints=150)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    e = slam.Extractor.from_params(params, cam)
    cap = params.max_nb_keypoints + e.grid_resolution[0] * e.grid_resolution[1] + 8
    sp = slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=np.tile([0.0, -6.3], (S, 1)))
    res = []
    for right in (r, r2):
        ks = slam.KeypointSet(S, cap)
        ks.detect(e, b)
        ks.stereo_match(b, right, params, sp, prior=2)
        res.append([ks.download(s) for s in range(S)])
        ks.close()
    for s in range(S):
        for k in ("yx", "ids", "stereo_yx", "has_stereo"):
            assert np.array_equal(res[0][s][k], res[1][s][k]), (s, k)
        assert res[0][s]["has_stereo"].sum() > 20
    ks = slam.KeypointSet(S, cap)
    ks.detect(e, b)
    with pytest.raises(Exception, match="TARGET_ONLY"):
        ks.flow_match(r2, b, params, sp, prior=2)
    ks.close()


def test_compute_pose_5pt_enqueue_only_filters_the_same_way(slam, syn):
    """P = status = NULL: the five-point seam only enqueues; the lists end up exactly as after the fetching call."""
    S, cap = 2, 400
    cam = syn.KITTI_CAM
    res = []
    for fetch in (True, False):
        ks = slam.KeypointSet(S, cap)
        for s in range(S):
            fs = syn.five_point_scene(n=250, seed=80 + s, noise_px=0.3, outlier_frac=0.25, iters=4)
            ks.upload(s, fs["px2"][:, ::-1], np.zeros(250, bool))
            ks.upload_keyframe(s, fs["px1"][:, ::-1], np.ones(250, bool))
        sp = slam.stream_params(S, Tcw=np.eye(4), cam=cam)
        r = ks.compute_pose_5pt(sp, iters=64, seed=3, fetch=fetch)
        assert (r is None) == (not fetch)
        res.append([ks.download(s)["ids"] for s in range(S)])
        ks.close()
    for s in range(S):
        assert np.array_equal(res[0][s], res[1][s]) and 150 < len(res[0][s]) < 250


def test_temporal_triangulation_on_the_set_equals_the_host_seam(slam, syn):
    """slam_kpset_triangulate_temporal (mapper.jl:185-262 on the device-resident lists: first observation + its key-frame id beside
    every keypoint) against slam_triangulate's temporal mode fed with numpy-gathered arrays, per observer key-frame: map points,
    is_3d flags and removals."""
    S, cap, nkf = 2, 500, 4
    cam = syn.KITTI_CAM
    dist = (0.0, 0.0, 0.0, 0.0)
    fx, fy, cx, cy = cam
    rng = np.random.default_rng(8)
    ks = slam.KeypointSet(S, cap)
    # key-frame poses (world -> camera): a camera moving sideways and forward, a little yaw
    def pose(k, s):
        th = 0.01 * k
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [-0.6 * k - 0.1 * s, 0.02 * k, -0.3 * k]
        return T
    kf_cw = np.stack([np.stack([pose(k, s) for k in range(nkf)]) for s in range(S)])
    kf_cur = np.array([3, 3], np.int32)
    Tcw_cur = np.stack([pose(3.4, s) for s in range(S)])             # the frame: a little past key-frame 3 (it IS key-frame 3 being created)
    Twc_cur = np.stack([np.linalg.inv(T) for T in Tcw_cur])
    proj = lambda T, X: (lambda Xc: np.stack([fy * Xc[:, 1] / Xc[:, 2] + cy, fx * Xc[:, 0] / Xc[:, 2] + cx], 1))((X @ T[:3, :3].T) + T[:3, 3])
    host = []
    for s in range(S):
        n = 300
        X = np.stack([rng.uniform(-15, 15, n), rng.uniform(-4, 4, n), rng.uniform(6, 60, n)], 1)
        fk = rng.integers(0, 4, n).astype(np.int32)                  # first observer 0..3 (3 = the frame itself: skipped)
        yx = proj(Tcw_cur[s], X) + rng.normal(0, 0.2, (n, 2))
        fyx = np.stack([proj(kf_cw[s, k], X[i:i + 1])[0] for i, k in enumerate(fk)]) + rng.normal(0, 0.2, (n, 2))
        bad = rng.random(n) < 0.15
        fyx[bad] += rng.uniform(25, 60, (int(bad.sum()), 2)) * rng.choice([-1, 1], (int(bad.sum()), 2))     # gross mismatches: gates + parallax
        is3 = rng.random(n) < 0.3
        haskf = rng.random(n) < 0.95
        ks.upload(s, yx, is3, np.where(is3[:, None], X, 0.0))
        ks.upload_keyframe(s, fyx, haskf)                             # (kyx is not used here; has_kf marks observed keypoints)
        ks.upload_first(s, fyx, fk, kf_count=4)
        host.append((yx, fyx, fk, is3, haskf))
    sp = slam.stream_params(S, cam=cam, dist=dist)
    before = [ks.download(s) for s in range(S)]
    ks.triangulate_temporal(sp, kf_cw, Twc_cur, kf_cur, kf_lo=np.array([1, 0], np.int32), max_error=3.0)
    after = [ks.download(s) for s in range(S)]
    cnt = ks.counts()
    n_new = 0
    for s in range(S):
        yx, fyx, fk, is3, haskf = host[s]
        keep = np.ones(len(yx), bool); new3 = is3.copy(); xyz = before[s]["xyz"].copy()
        lo = [1, 0][s]
        for k in range(nkf):
            m = ~is3 & haskf & (fk == k) & (fk != kf_cur[s]) & (fk >= lo)
            if not m.any():
                continue
            rel = kf_cw[s, k] @ Twc_cur[s]; rel_inv = np.linalg.inv(rel)
            _, p2, _ = slam.pose_inputs(cam, dist, yx[m], np.zeros((int(m.sum()), 3)))          # undistorted (x, y)
            _, p1, _ = slam.pose_inputs(cam, dist, fyx[m], np.zeros((int(m.sum()), 3)))
            b = np.stack([(p2[:, 0] - cx) / fx, (p2[:, 1] - cy) / fy, np.ones(len(p2))], 1) @ rel[:3, :3].T
            par = np.linalg.norm(p1[:, ::-1] - np.stack([fy * b[:, 1] / b[:, 2] + cy, fx * b[:, 0] / b[:, 2] + cx], 1), axis=1)
            Xc, ok = slam.triangulate(cam, cam, rel_inv, p1[:, ::-1], p2[:, ::-1], 3.0, parallax=par, min_parallax=20.0)
            idx = np.flatnonzero(m)
            keep[idx[~ok]] = False
            new3[idx[ok]] = True
            Wob = np.linalg.inv(kf_cw[s, k])
            xyz[idx[ok]] = Xc[ok] @ Wob[:3, :3].T + Wob[:3, 3]
            n_new += int(ok.sum())
        assert cnt[s] == keep.sum() == len(after[s]["yx"]), (s, cnt[s], keep.sum())
        assert np.array_equal(after[s]["ids"], before[s]["ids"][keep]), s
        assert np.array_equal(after[s]["is_3d"].astype(bool), new3[keep]), s
        got = after[s]["xyz"][new3[keep]]; want = xyz[keep][new3[keep]]
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9), (s, np.abs(got - want).max())
    assert n_new > 150                                                 # the call did triangulate, and removed some
    assert sum(len(b["yx"]) for b in before) - int(cnt.sum()) > 10
    # slam_kpset_keyframe: fresh keypoints get their first observation and the running key-frame id; old ones keep theirs
    f0, k0, c0 = ks.download_first(0)
    ks.keyframe()
    f1, k1, c1 = ks.download_first(0)
    assert c1 == c0 + 1 == 5
    hk = ks.download_keyframe(0)[1]
    assert hk.all()
    was = after[0]
    old = np.isin(was["ids"], before[0]["ids"][host[0][4]])           # had a key-frame observation before the call
    assert np.array_equal(k1[old], k0[old]) and np.array_equal(f1[old], f0[old])
    assert (k1[~old] == 4).all() and np.array_equal(f1[~old], was["yx"][~old])


def test_a_small_n_bound_is_only_a_hint(slam, orc, syn, texture):
    """n_bound sizes the launches from a host-side estimate; an estimate below the live count must not leave keypoints
    with a stale status -- the kernels walk the work list in rounds.  Same lists with n_bound = 64 as with the default bound."""
    S, H, W = 3, 120, 160
    streams, a, b, r, keep = _setup(slam, texture, S, H, W)
    params = slam.Params(stereo=True, max_nb_keypoints=150)
    rng = np.random.default_rng(9)
    out = []
    for bound in (0, 64):
        ks = slam.KeypointSet(S, 200)
        for s in range(S):
            k = orc.detect(streams[s][0][0], np.zeros((0, 2)), max_points=150).astype(float)
            ks.upload(s, k, np.arange(len(k)) % 3 == 0)
        shift = np.array([streams[s][2][1] for s in range(S)])
        ks.flow_match(a, b, params, slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=shift), prior=2, n_bound=bound)
        ks.stereo_match(b, r, params, slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=np.tile([0.0, -6.3], (S, 1))), prior=2, n_bound=bound)
        T21 = np.eye(4); T21[0, 3] = -0.54
        ks.triangulate(syn.KITTI_CAM, syn.KITTI_CAM, T21, np.eye(4), max_error=3.0, n_bound=bound)
        out.append([ks.download(s) for s in range(S)])
        assert ks.counts().sum() > 3 * 64
        ks.close()
    for s in range(S):
        for key in ("yx", "is_3d", "xyz", "stereo_yx", "has_stereo"):
            m = out[0][s]["is_3d"] if key == "xyz" else (out[0][s]["has_stereo"] if key == "stereo_yx" else slice(None))
            assert np.array_equal(out[0][s][key][m], out[1][s][key][m]), (s, key)


def test_more_than_64_streams(slam, orc, syn, texture):
    """Up to 128 streams share a keypoint set and its launches (bench.py's default since the end of round 3: 128): the work list's
    offsets are sums over MORE than one wave's worth of stream counts.  70 small streams with ragged lists: temporal match + detect +
    stereo match on the set; streams 0, 63, 64 and 69 are replayed through the oracle."""
    S, H, W = 70, 96, 128
    streams, a, b, r, keep = _setup(slam, texture, S, H, W)
    params = slam.Params(stereo=True, max_nb_keypoints=80)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    e = slam.Extractor.from_params(params, cam)
    ncell = e.grid_resolution[0] * e.grid_resolution[1]
    ks = slam.KeypointSet(S, params.max_nb_keypoints + ncell + 8)
    kps, is3 = [], []
    for s in range(S):
        k = orc.detect(streams[s][0][0], np.zeros((0, 2)), max_points=30 + (s * 7) % 40).astype(float)
        kps.append(k); is3.append(np.arange(len(k)) % 2 == 0)
        ks.upload(s, k, is3[s])
    shift = np.array([streams[s][2][1] for s in range(S)])
    ks.flow_match(a, b, params, slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=shift), prior=2)
    cnt = ks.counts()
    check = (0, 63, 64, 69)
    lists = {}
    f = lambda im: np.asfortranarray(im)
    for s in check:
        ra, rb = orc.pyr_build(f(streams[s][0][0]), 3, 1.0, 1), orc.pyr_build(f(streams[s][0][1]), 3, 1.0, 1)
        ref = orc.optical_flow_matching(ra, rb, kps[s], is3[s], kps[s] + shift[s], (H, W), sum_order=1)
        keep_ = ~ref["removed"]
        got = ks.download(s)
        assert cnt[s] == keep_.sum() == len(got["yx"]), s
        assert np.abs(got["yx"] - ref["new_pixels"][keep_]).max() <= 1e-9, s
        lists[s] = got["yx"]
    ks.detect(e, b)
    cnt2 = ks.counts()
    for s in check:
        fresh = orc.detect(f(streams[s][0][1]), lists[s], max_points=params.max_nb_keypoints).astype(float)
        got = ks.download(s)
        assert cnt2[s] == len(lists[s]) + len(fresh), s
        assert np.array_equal(got["yx"][len(lists[s]):], fresh), s
    ks.close()
