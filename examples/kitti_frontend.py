#!/usr/bin/env python3
"""Front-end hot path over a KITTI odometry sequence, frame by frame, through the seams the reference's run loop
uses (example/kitty/main.jl:8-62 feeds the frames; front_end.jl:454-477 / map_manager.jl:98-113, 451-564 /
mapper.jl:51-66, 142-183 consume them):

    8-bit PNG -> slam_pyr_update_u8 (left)  -> optical_flow_matching! of the tracked keypoints
    key-frames: detect -> right pyramid -> stereo optical_flow_matching! -> triangulate_stereo!

It is an array-level driver, not SLAM: no map, no pose graph -- it reports what the seams produce (tracked keypoints,
stereo matches, triangulated depths, device time per stage) so that a maintainer can compare against the reference's
@debug timers on the same sequence.

    python examples/kitti_frontend.py /data/kitti/dataset 05 --frames 200
    python examples/kitti_frontend.py --synthetic --frames 12        # no dataset at hand: writes a small KITTI-shaped one
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slam_jl_amd as slam  # noqa: E402
from slam_jl_amd import kitti, synthetic as syn  # noqa: E402


def run(dataset, n_frames=None, kf_every=5, max_keypoints=1000, ctx=None, verbose=False):
    """Returns a list of per-frame dicts: frame, keyframe, tracked, detected, stereo, depth_median, ms."""
    ctx = ctx or slam.default_context(0)
    n_frames = min(n_frames or len(dataset), len(dataset))
    left0, _ = dataset[0]
    H, W = left0.shape
    fx, fy, cx, cy = dataset.intrinsics
    params = slam.Params(stereo=dataset.stereo, max_nb_keypoints=max_keypoints)
    cam = slam.Camera(fx, fy, cx, cy, height=H, width=W)
    ex = slam.Extractor.from_params(params, cam)
    prev = slam.LKPyramid(shape=(H, W), levels=params.pyramid_levels, ctx=ctx)
    cur = slam.LKPyramid(shape=(H, W), levels=params.pyramid_levels, ctx=ctx)
    rpyr = slam.LKPyramid(shape=(H, W), levels=params.pyramid_levels, ctx=ctx)
    kp = np.zeros((0, 2)); is3d = np.zeros(0, dtype=bool)
    stats = []
    for i in range(n_frames):
        left, right = dataset[i]
        t0 = time.perf_counter()
        prev, cur = cur, prev                                             # preprocess!: pyramid swap + update!
        slam.update_(cur, left, sigma=params.pyramid_sigma)               # uint8 in, converted on the device
        tracked = 0
        if i > 0 and len(kp):
            new, ok = slam.optical_flow_matching(prev, cur, kp, is3d, kp, params, ctx=ctx)
            kp, is3d = new[ok], is3d[ok]
            tracked = int(ok.sum())
        row = dict(frame=i, keyframe=False, tracked=tracked, detected=0, stereo=0, depth_median=float("nan"))
        if i % kf_every == 0:                                             # create_keyframe! + mapper's stereo path
            fresh = slam.detect(ex, cur, kp, ctx=ctx).astype(np.float64)
            kp = np.concatenate([kp, fresh]); is3d = np.concatenate([is3d, np.zeros(len(fresh), dtype=bool)])
            row.update(keyframe=True, detected=len(fresh))
            if dataset.stereo and len(kp):
                slam.update_(rpyr, right, sigma=params.pyramid_sigma)
                rk, ok = slam.optical_flow_matching(cur, rpyr, kp, np.zeros(len(kp), dtype=bool), kp, params, ctx=ctx)
                ok &= np.abs(rk[:, 0] - kp[:, 0]) < 2.0                    # epipolar check of stereo_matching!, rectified pair
                if ok.any():
                    camt = (fx, fy, cx, cy)
                    xyz, good = slam.triangulate(camt, camt, dataset.Ti0, kp[ok], rk[ok], params.max_reprojection_error, ctx=ctx)
                    idx = np.where(ok)[0][good]
                    is3d[idx] = True
                    row.update(stereo=int(good.sum()), depth_median=float(np.median(xyz[good, 2])) if good.any() else float("nan"))
        row["ms"] = (time.perf_counter() - t0) * 1e3
        stats.append(row)
        if verbose:
            print(row)
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("base_dir", nargs="?")
    ap.add_argument("sequence", nargs="?", default="05")
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--mono", action="store_true")
    ap.add_argument("--synthetic", action="store_true", help="write a small KITTI-shaped sequence to a temp dir and run on it")
    args = ap.parse_args()
    tmp = None
    if args.synthetic or not args.base_dir:
        tmp = tempfile.TemporaryDirectory()
        L, R, _ = syn.stereo_stream("kitti05", args.frames, seed=0, disparity=12.4)
        kitti.write_synthetic_sequence(tmp.name, args.sequence, L, R, syn.KITTI_CAM, 0.54)
        args.base_dir = tmp.name
    ds = slam.KittyDataset(args.base_dir, args.sequence, stereo=not args.mono)
    print(ds)
    st = run(ds, args.frames, verbose=True)
    ms = np.array([r["ms"] for r in st[1:]])
    print(f"{len(st)} frames, median {np.median(ms):.2f} ms/frame wall incl. PNG-decoded u8 upload ({1e3 / np.median(ms):.0f} frames/s single stream)")
    if tmp:
        tmp.cleanup()


if __name__ == "__main__":
    main()
