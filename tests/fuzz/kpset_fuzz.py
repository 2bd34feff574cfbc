"""Random key-frame steps on the device-resident keypoint lists (slam_kpset_*: temporal match, cull, detect + merge, stereo match,
triangulate) against the host protocol on the batch seams and the oracle's optical_flow_matching! -- tests/test_gpu_kpset.py's step
with random stream counts, shapes, list sizes (empty streams, full lists, everything culled): python tests/fuzz/kpset_fuzz.py [n] [seed0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
from oracle import oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30; s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0


class Bad(Exception):
    pass


def need(cond, msg):
    if not cond: raise Bad(msg)


def step(seed):
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 7)); H = int(rng.integers(80, 200)); W = int(rng.integers(100, 260)); levels = 3 if min(H, W) >= 64 else 2
    maxkp = int(rng.integers(20, 300)); cull_p = float(rng.choice([0.0, 0.2, 0.6])); disp = float(rng.uniform(3, 9))
    streams = [syn.stereo_stream((H, W), 2, 30 + s + 7 * seed, (1.0 + 0.2 * s, -1.4), disp) for s in range(S)]
    a = slam.PyramidBatch((H, W), levels=levels, S=S); b = slam.PyramidBatch((H, W), levels=levels, S=S); r = slam.PyramidBatch((H, W), levels=levels, S=S)
    dev = lambda k, f: [torch.from_numpy(np.ascontiguousarray(st[k][f].T)).cuda() for st in streams]
    d0, d1, dr = dev(0, 0), dev(0, 1), dev(1, 1)
    torch.cuda.synchronize()
    a.update_([d.data_ptr() for d in d0]); b.update_([d.data_ptr() for d in d1]); r.update_([d.data_ptr() for d in dr])
    params = slam.Params(stereo=True, max_nb_keypoints=maxkp, pyramid_levels=levels)
    cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
    e = slam.Extractor.from_params(params, cam)
    ncell = e.grid_resolution[0] * e.grid_resolution[1]
    cap = maxkp + ncell + 8
    kps, is3, sid = [], [], []
    for s in range(S):
        n0 = int(rng.choice([0, 1, 5, maxkp // 2, maxkp, maxkp + 5]))
        k = orc.detect(streams[s][0][0], np.zeros((0, 2)), max_points=max(n0, 1)).astype(float)[:n0]
        if n0 and rng.random() < 0.5: k = np.concatenate([k, np.array([[1.0, 1.0], [H, W], [2.5, W - 1.5]])])
        k = k[:cap]
        kps.append(k.reshape(-1, 2)); is3.append(rng.random(len(k)) < 0.5); sid.append(np.full(len(k), s, np.int32))
    ks = slam.KeypointSet(S, cap)
    for s in range(S): ks.upload(s, kps[s], is3[s])
    shift = np.array([streams[s][2][1] for s in range(S)])
    sp = slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=shift)
    tag = f"S {S} {H}x{W} max_kp {maxkp} cull {cull_p} n0 {[len(k) for k in kps]}"
    # ---- temporal match ----
    ks.flow_match(a, b, params, sp, prior=2)
    P = np.concatenate(kps); T = np.concatenate(is3); I = np.concatenate(sid)
    proj = P + shift[I] if len(P) else P
    inside = (proj[:, 0] >= 1) & (proj[:, 0] <= H) & (proj[:, 1] >= 1) & (proj[:, 1] <= W)
    skip = T & ~inside
    pos = np.full((len(P), 2), np.nan); alive = np.zeros(len(P), bool)
    if (~skip).any():
        hk, h3, hs, hsrc = slam.optical_flow_matching_batch_kept(a, b, I[~skip], P[~skip], T[~skip], proj[~skip], params)
        idx_ns = np.flatnonzero(~skip)
        pos[idx_ns[hsrc]] = hk; alive[idx_ns[hsrc]] = True
    pos[skip] = P[skip]; alive[skip] = True
    cnt = ks.counts()
    for s in range(S):
        m = (I == s) & alive
        got = ks.download(s)
        need(cnt[s] == m.sum() == len(got["yx"]), f"{tag}: temporal count stream {s}: {cnt[s]} vs {m.sum()}")
        need(np.array_equal(got["yx"], pos[m]) and np.array_equal(got["is_3d"], T[m]), f"{tag}: temporal lists stream {s}")
        if len(kps[s]):
            ra, rb = orc.pyr_build(streams[s][0][0], levels, 1.0, 1), orc.pyr_build(streams[s][0][1], levels, 1.0, 1)
            ref = orc.optical_flow_matching(ra, rb, kps[s], is3[s], kps[s] + shift[s], (H, W), sum_order=1, pyramid_levels=levels)
            keep_ref = ~ref["removed"]
            need(np.array_equal(keep_ref, alive[I == s]), f"{tag}: temporal vs oracle, survivors of stream {s}")
            if keep_ref.any(): need(np.abs(got["yx"] - ref["new_pixels"][keep_ref]).max() <= 1e-9, f"{tag}: temporal vs oracle, positions of stream {s}")
    P, T, I = pos[alive], T[alive], I[alive]
    # ---- cull ----
    flags = np.zeros((S, cap), np.uint8)
    for s in range(S):
        k = int(cnt[s]); flags[s, :k] = rng.random(k) < (1.0 if (cull_p > 0 and s == 0) else cull_p)      # stream 0: everything culled
    fdev = torch.from_numpy(flags).cuda(); torch.cuda.synchronize()
    ks.remove(fdev.data_ptr())
    keepm = np.concatenate([flags[s, :int(cnt[s])] == 0 for s in range(S)]) if len(P) else np.zeros(0, bool)
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
        need(cnt2[s] == len(lists[s][0]), f"{tag}: detect count stream {s}: {cnt2[s]} vs {len(lists[s][0])}")
        need(np.array_equal(got["yx"], lists[s][0]) and np.array_equal(got["is_3d"], lists[s][1]), f"{tag}: detect lists stream {s}")
        need(len(np.unique(got["ids"])) == len(got["ids"]), f"{tag}: ids not unique, stream {s}")
    # ---- stereo match ----
    sps = slam.stream_params(S, cam=syn.KITTI_CAM, shift_yx=np.tile([0.0, -disp], (S, 1)))
    ks.stereo_match(b, r, params, sps, prior=2)
    for s in range(S):
        kp, t3 = lists[s]
        got = ks.download(s)
        if len(kp) == 0:
            need(len(got["yx"]) == 0, f"{tag}: stereo, empty stream {s}"); lists[s] = (kp, t3, np.zeros((0, 2)), np.zeros(0, bool)); continue
        res = slam.optical_flow_matching_frame(b.pyramids[s], r.pyramids[s], kp, t3, kp + np.array([0.0, -disp]), params, (H, W), stereo=True,
                                               undistorted_left=kp, right_cam=syn.KITTI_CAM)
        keep_s = ~res["removed"]
        need(np.array_equal(got["yx"], kp[keep_s]), f"{tag}: stereo positions stream {s}")
        need(np.array_equal(got["has_stereo"], res["updated"][keep_s]), f"{tag}: stereo flags stream {s}")
        up = got["has_stereo"]
        need(np.array_equal(got["stereo_yx"][up], res["new_pixels"][keep_s][up]), f"{tag}: stereo pixels stream {s}")
        lists[s] = (kp[keep_s], t3[keep_s], got["stereo_yx"], up)
    # ---- triangulation ----
    T21 = np.eye(4); T21[0, 3] = -0.54
    Twc = np.eye(4); Twc[:3, 3] = [1.0, 2.0, 3.0]
    ks.triangulate(syn.KITTI_CAM, syn.KITTI_CAM, T21, Twc, max_error=3.0)
    for s in range(S):
        kp, t3, syx, up = lists[s]
        got = ks.download(s)
        cand = up & ~t3
        if not cand.any():
            need(np.array_equal(got["is_3d"], t3), f"{tag}: triangulate without candidates, stream {s}"); continue
        xyz, ok = slam.triangulate(syn.KITTI_CAM, syn.KITTI_CAM, T21, kp[cand], syx[cand], 3.0)
        exp3 = t3.copy(); exp3[np.flatnonzero(cand)[ok]] = True
        exps = up.copy(); exps[np.flatnonzero(cand)[~ok]] = False
        need(np.array_equal(got["is_3d"], exp3) and np.array_equal(got["has_stereo"], exps), f"{tag}: triangulate flags stream {s}")
        if ok.any():
            world = xyz[ok] + Twc[:3, 3]
            need(np.abs(got["xyz"][np.flatnonzero(cand)[ok]] - world).max() <= 1e-9 * max(1.0, np.abs(world).max()), f"{tag}: triangulated points stream {s}")
    return tag


for seed in range(s0, s0 + n):
    try:
        step(seed)
    except Bad as ex:
        fails += 1; print("FAIL seed", seed, ex, flush=True)
    except Exception as ex:
        import traceback
        fails += 1; print("FAIL seed", seed, "exception", repr(ex)[:300], traceback.format_exc().splitlines()[-3][:200], flush=True)
print(f"{n} key-frame steps, {fails} failures")
