#!/usr/bin/env python3
"""The reference's per-frame front-end (front_end.jl:60-113 + the mapper's stereo path, mapper.jl:51-66, 142-183) for S
lock-stepped stereo streams with every keypoint list resident in HBM -- the calls a host issues per frame:

    left frames  -> PyramidBatch.update_                                   (preprocess!)
    KeypointSet.flow_match      prior = projection of the map points under the predicted pose   (klt_tracking!)
    KeypointSet.compute_pose_5pt   epipolar outlier filter against the previous key-frame       (compute_pose_5pt!)
    KeypointSet.compute_pose       P3P RANSAC + PnP refinement -> the frame's pose              (compute_pose!)
    key-frames: detect -> keyframe -> right frames -> stereo_match -> triangulate -> triangulate_temporal   (create_keyframe!, mapper)

No host keypoint arrays anywhere; per frame the host receives S poses, S status words and S list lengths.  It is an
array-level driver, not SLAM (no map maintenance, no bundle adjustment, no relocalisation): what it shows is that the seams
compose -- on a rigid synthetic scene the poses it returns are the camera motion.

    python examples/device_frontend.py --frames 12 --streams 4
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slam_jl_amd as slam  # noqa: E402
from slam_jl_amd import synthetic as syn  # noqa: E402


def run(lefts, rights, cam, baseline, kf_every=4, max_keypoints=300, seed=0, ctx=None, verbose=False):
    """lefts / rights: per stream a list of float images in [0, 1] (H, W).  Returns per frame a dict with the S poses
    (world -> camera, world = the first frame's camera), status words, list lengths and the wall time."""
    import torch
    ctx = ctx or slam.default_context(0)
    S = len(lefts); n_frames = len(lefts[0])
    H, W = lefts[0][0].shape
    fx, fy, cx, cy = cam
    params = slam.Params(stereo=True, max_nb_keypoints=max_keypoints)
    camera = slam.Camera(fx, fy, cx, cy, height=H, width=W)
    ex = slam.Extractor.from_params(params, camera)
    ncell = ex.grid_resolution[0] * ex.grid_resolution[1]
    ks = slam.KeypointSet(S, max_keypoints + ncell + 8, ctx=ctx)
    prev = slam.PyramidBatch((H, W), levels=params.pyramid_levels, S=S, ctx=ctx)
    cur = slam.PyramidBatch((H, W), levels=params.pyramid_levels, S=S, ctx=ctx)
    rpyr = slam.PyramidBatch((H, W), levels=params.pyramid_levels, S=S, ctx=ctx)
    dev = lambda im: torch.from_numpy(np.ascontiguousarray(im.T)).cuda()          # Julia layout: column-major H x W
    T21 = np.eye(4); T21[0, 3] = -baseline                                          # left camera -> right camera
    Tcw = np.tile(np.eye(4), (S, 1, 1)); Tkf = Tcw.copy()
    NKF = 8; kf_cw = np.tile(np.eye(4), (S, NKF, 1, 1)); n_kf = 0          # poses of the last key-frames (triangulate_temporal!'s observers)
    sp_right = slam.stream_params(S, cam=cam, shift_yx=np.zeros((S, 2)))
    out = []
    for i in range(n_frames):
        t0 = time.perf_counter()
        prev, cur = cur, prev
        frames = [dev(lefts[s][i]) for s in range(S)]
        torch.cuda.synchronize()
        cur.update_([f.data_ptr() for f in frames], sigma=params.pyramid_sigma, ctx=ctx)
        status = np.zeros(S, np.int32); st5 = np.zeros(S, np.int32)
        if i > 0:
            # motion model: the last pose (constant-position prediction is enough for the prior of a 3-D keypoint)
            ks.flow_match(prev, cur, params, slam.stream_params(S, Tcw=Tcw, cam=cam), prior=1, ctx=ctx)
            # R_compensation = Rcw(key-frame) * Rwc(frame), front_end.jl:251
            Rc = np.tile(np.eye(4), (S, 1, 1))
            for s in range(S):
                Rc[s, :3, :3] = Tkf[s, :3, :3] @ Tcw[s, :3, :3].T
            Rt5, st5, n5, par, _ = ks.compute_pose_5pt(slam.stream_params(S, Tcw=Rc, cam=cam), min_parallax=5.0,
                                                       max_repr_error=params.max_reprojection_error, iters=64, seed=seed + 2 * i, ctx=ctx)
            poses, status, ninl, _ = ks.compute_pose(slam.stream_params(S, cam=cam), threshold=params.max_reprojection_error,
                                                     iters=128, seed=seed + 2 * i + 1, ctx=ctx)
            for s in range(S):
                if status[s]:
                    Tcw[s] = poses[s]
        if i % kf_every == 0:
            ks.detect(ex, cur, ctx=ctx)
            ks.keyframe(ctx=ctx)
            Tkf = Tcw.copy()
            kfid = n_kf; kf_cw[:, kfid % NKF] = Tcw; n_kf += 1
            rframes = [dev(rights[s][i]) for s in range(S)]
            torch.cuda.synchronize()
            rpyr.update_([f.data_ptr() for f in rframes], sigma=params.pyramid_sigma, ctx=ctx, target_only=True)   # only matched INTO
            ks.stereo_match(cur, rpyr, params, sp_right, prior=2, ctx=ctx)
            Twc = np.stack([np.linalg.inv(Tcw[s]) for s in range(S)])
            ks.triangulate(cam, cam, T21, Twc, max_error=params.max_reprojection_error, ctx=ctx)
            if kfid > 0:                                                    # mapper.jl:86: 2-D keypoints left over, against their first observers
                ks.triangulate_temporal(slam.stream_params(S, cam=cam), kf_cw, Twc, kfid, max_error=params.max_reprojection_error, ctx=ctx)
        cnt = ks.counts(ctx=ctx)
        row = dict(frame=i, keyframe=i % kf_every == 0, poses=Tcw.copy(), status=status.copy(), status_5pt=np.asarray(st5).copy(),
                   counts=cnt.copy(), ms=(time.perf_counter() - t0) * 1e3)
        out.append(row)
        if verbose:
            print(f"frame {i:3d} kf={row['keyframe']!s:5} keypoints {cnt.tolist()} pose ok {status.tolist()} 5pt ok {np.asarray(st5).tolist()} "
                  f"t = {np.round(Tcw[0][:3, 3], 3).tolist()} {row['ms']:.2f} ms")
    n3 = [int(ks.download(s)["is_3d"].sum()) for s in range(S)]
    ks.close()
    return out, n3


def synthetic_scene(S, n_frames, shape=(200, 320), disparity=8.0, seed=0):
    """A fronto-parallel textured plane seen by cameras translating parallel to it: per stream left / right frames, the
    per-frame image offsets (y, x) in pixels, and the plane's depth Z = fx b / d (the image shift of o pixels is the camera
    translation -o Z / f)."""
    lefts, rights, offs = [], [], []
    for s in range(S):
        L, R, flows = syn.stereo_stream(shape, n_frames, seed=seed + s, step=(1.1 + 0.3 * s, -1.6 + 0.2 * s), disparity=disparity)
        lefts.append(L); rights.append(R); offs.append(np.asarray(flows, dtype=np.float64))
    return lefts, rights, offs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--replay-dir", default=None, help="write stream s's camera positions to DIR/stream_s as a ReplaySaver dump (src/io/saver.jl)")
    args = ap.parse_args()
    cam, baseline, disparity = syn.KITTI_CAM, 0.54, 8.0
    lefts, rights, offs = synthetic_scene(args.streams, args.frames, disparity=disparity)
    out, n3 = run(lefts, rights, cam, baseline, verbose=True)
    Z = cam[0] * baseline / disparity
    for s in range(args.streams):
        o = offs[s][-1] - offs[s][0]
        want = np.array([o[1] * Z / cam[0], o[0] * Z / cam[1], 0.0])
        got = out[-1]["poses"][s][:3, 3]
        print(f"stream {s}: translation {np.round(got, 3).tolist()} m, expected {np.round(want, 3).tolist()} m (plane at {Z:.1f} m), {n3[s]} map points")
    if args.replay_dir:
        import os
        for s in range(args.streams):
            saver = slam.ReplaySaver()                      # set_frame_wc!(saver, frame id, wc) as SlamManager does after every frame
            for r in out:
                saver.set_frame_wc(r["frame"], np.linalg.inv(r["poses"][s]))
            saver.save(os.path.join(args.replay_dir, f"stream_{s}"))
        print(f"wrote positions.bson / ids.bson for {args.streams} streams under {args.replay_dir}")
    ms = np.array([r["ms"] for r in out[1:]])
    print(f"median {np.median(ms):.2f} ms per step of {args.streams} frames (wall, incl. the host -> device frame copies)")


if __name__ == "__main__":
    main()
