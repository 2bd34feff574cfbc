"""GPU: the monocular pose loop of bench.py's configs[3] leg (benchlib/lockstep.py, pose=True / stereo=False: klt_tracking! with the
motion model's prior, compute_pose_5pt!, compute_pose!, key-frames with triangulate_temporal! -- front_end.jl:132-219, mapper.jl:185-262)
recovers the camera motion of the rigid synthetic scene: every compute_pose! accepted, translation within 5 cm of the frames' offsets,
map points on the scene plane.

Round 4's bench reported 162 m here at S = 128: k_kpset_kf_advance ran one 64-thread block, so the key-frame counters of streams 64..
never moved, every later detection of those streams was stamped "first observed by key-frame 0" and triangulate_temporal! paired it with
its own position under key-frame 0's pose -- zero parallax, points at +-1e8 m.  S = 72 covers streams beyond 64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,periods", [(8, 6), (72, 4)])
def test_mono_loop_recovers_the_translation(slam, syn, S, periods):
    import torch
    from benchlib.lockstep import run_lockstep_kpset, make_workload
    wl = make_workload(slam, syn, "euroc_mono", seed=0, streams=S)
    camt, Z = wl["camt"], 30.0
    worst = {"t": 0.0, "map": 0.0, "far": 0}

    def diag(i, kf, pst, ks, ctx, off_now):
        if pst["ref"] is None:
            return
        off = off_now - pst["ref"]
        want = np.stack([off[:, 1] * Z / camt[0], off[:, 0] * Z / camt[1], np.zeros(S)], axis=1)
        worst["t"] = max(worst["t"], float(np.abs(pst["Tcw"][:, :3, 3] - want).max()))
        if kf:                                                      # the map points of a few streams (incl. the last) against the scene plane
            for s in sorted({0, S // 2, S - 1}):
                d = ks.download(s, ctx=ctx)
                m = d["is_3d"]
                ref_px = d["yx"][m] - off[s]
                Xw = np.stack([(ref_px[:, 1] - camt[2]) / camt[0] * Z, (ref_px[:, 0] - camt[3]) / camt[1] * Z, np.full(int(m.sum()), Z)], axis=1)
                e = np.abs(d["xyz"][m] - Xw).max(axis=1)
                worst["map"] = max(worst["map"], float(np.median(e))); worst["far"] = max(worst["far"], int((e > 3.0).sum()))

    r = run_lockstep_kpset(slam, torch, 0, wl, periods, 2, 1, None, torch.device("cuda", 0), "host_u8", pose=True, diag=diag)
    p = r["pose"]
    assert p["accepted_fraction"] == 1.0, p                          # every compute_pose! accepted
    assert p["max_translation_error_m"] < 0.05, p                    # within 5 cm of the ground truth (timed region)
    assert worst["t"] < 0.05, worst                                  # ... and in the warm-up frames
    assert worst["map"] < 0.25 and worst["far"] == 0, worst          # temporal triangulation: no point off the plane by metres
    assert p["pose_ok"]
    assert r["tracked_kpts_per_frame"] > 500


def test_keyframe_counter_advances_for_every_stream(slam):
    """slam_kpset_keyframe stamps new keypoints with the stream's key-frame counter and advances it -- for all S <= 128 streams"""
    S, cap = 128, 16
    ks = slam.KeypointSet(S, cap)
    for s in range(S):
        ks.upload(s, np.array([[10.0 + s, 20.0]]), np.zeros(1, bool))
    ks.keyframe(); ks.keyframe()
    for s in (0, 63, 64, 100, 127):
        fyx, fkf, kc = ks.download_first(s)
        assert kc == 2 and fkf[0] == 0 and np.array_equal(fyx[0], [10.0 + s, 20.0]), (s, kc, fkf)
        ks.upload(s, np.array([[10.0 + s, 20.0], [5.0, 6.0]]), np.zeros(2, bool))       # a fresh list: haskf cleared
    ks.keyframe()
    for s in (0, 63, 64, 100, 127):
        _, fkf, kc = ks.download_first(s)
        assert kc == 3 and list(fkf) == [2, 2], (s, kc, fkf)
    ks.close()
