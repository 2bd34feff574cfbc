"""Checker side of bench.py's cpu_baseline leg: the oracle's replay of a recorded run of the headline loop.  The oracle module is passed
in by the caller (bench.py imports it in that leg only); nothing here imports it."""
import numpy as np


def replay_stream_on_oracle(orc, slam, wl, rec, res, s, threads, with_ids=False):
    """The oracle's replay of stream s of a recorded run_lockstep_kpset run (checker; cpu_baseline leg only): the same 8-bit
    frames, prior shifts and cull flags through orc.pyr_build / optical_flow_matching / detect / triangulate
    (pyramid.jl:81-137, map_manager.jl:451-564 + :579-590, extractor.jl:63-95, mapper.jl:142-183).  Returns (yx, is_3d) -- with_ids: (yx, is_3d, ids), the
    keypoint ids counted in creation order as the device-resident lists count them."""
    from slam_jl_amd.triangulation import projection_matrices
    H, W, e, camt, disparity = wl["H"], wl["W"], wl["extractor"], wl["camt"], wl["disparity"]
    seq, period = res["seq"], res["period"]
    u8f = lambda im: np.asfortranarray(np.round(im * 255).astype(np.uint8).astype(np.float64) / 255.0)
    T21 = np.eye(4); T21[0, 3] = -(disparity * 30.0 / camt[0])
    P1, P2 = projection_matrices(camt, camt, T21)
    kp = np.zeros((0, 2)); is3 = np.zeros(0, bool); ids = np.zeros(0, np.int64); next_id = 0
    prev = None
    for r in rec["steps"]:
        i = r["i"]
        f = seq[(i % period) + s]
        img = u8f(wl["left"][f])
        cur = orc.pyr_build(img, wl["levels"], 1.0, 1)
        if len(kp) and r["shift"] is not None:
            ref = orc.optical_flow_matching(prev, cur, kp, is3, kp + r["shift"][s], (H, W), sum_order=1, threads=threads)
            keep = ~ref["removed"]
            kp, is3, ids = ref["new_pixels"][keep], is3[keep], ids[keep]
        if r["kf"]:
            keep = r["cull"][s, :len(kp)] == 0
            kp, is3, ids = kp[keep], is3[keep], ids[keep]
            fresh = orc.detect(img, kp, max_points=e.max_points, radius=e.radius, cell_size=e.cell_size).astype(np.float64)
            kp = np.concatenate([kp, fresh]); is3 = np.concatenate([is3, np.zeros(len(fresh), bool)])
            ids = np.concatenate([ids, next_id + np.arange(len(fresh), dtype=np.int64)]); next_id += len(fresh)
            rp = orc.pyr_build(u8f(wl["right"][f]), wl["levels"], 1.0, 1)
            ref = orc.optical_flow_matching(cur, rp, kp, is3, kp + np.array([0.0, -disparity]), (H, W), stereo=True, undistorted_left=kp,
                                            right_cam=camt, sum_order=1, threads=threads)
            keep = ~ref["removed"]
            kp, is3, ids = kp[keep], is3[keep], ids[keep]
            up, syx = ref["updated"][keep], ref["new_pixels"][keep]
            cand = np.flatnonzero(up & ~is3)
            if len(cand):
                _, ok = orc.triangulate(P1, P2, T21, camt, camt, kp[cand], syx[cand], 3.0)
                is3 = is3.copy(); is3[cand[ok]] = True
        prev = cur
    return (kp, is3, ids) if with_ids else (kp, is3)


