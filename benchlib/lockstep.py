"""The lock-stepped legs of bench.py: S streams per GPU sharing every launch (run_lockstep: host keypoint lists, the round-1 protocol;
run_lockstep_kpset: device-resident keypoint sets, the headline), the stage spans of one batched build, and the workloads BASELINE names."""
import os
import time

import numpy as np

from .common import (KF_EVERY, RIGHT_TARGET_ONLY, CULL_FRACTION, HBM_PEAK_GBS, HBM_ACHIEVABLE_GBS, pyramid_bytes, iir_rows_bytes,
                     frame_sequence, frame_sequence_n)


def run_lockstep(slam, torch, local_rank, S, steps, warmup, H, W, left_dev, right_dev, flows, disparity, params, extractor, fast, world, dist, dev, hook=None):
    """S streams in lock-step through the batch entry points.  Stream s plays the same ping-pong sequence shifted
    by s frames (so the S images of a step differ); key-frames fall on the same step for all streams."""
    # tracking stream in a scheduling class of its own, as in run_lockstep_kpset (hardware-queue aliasing with the pyramid graph's branches)
    ctx, ctx_pyr, ctx_right = (leg_ctx(slam, local_rank, int(os.environ.get("SLAM_BENCH_TRACK_PRIO", "-1"))), leg_ctx(slam, local_rank),
                               leg_ctx(slam, local_rank))
    levels = params.pyramid_levels
    AHEAD = max(1, int(os.environ.get("SLAM_BENCH_AHEAD", "1")))   # pyramid builds kept in flight ahead of the step being tracked (2 measured 5 % slower: two builds + LK contend for the HBM)
    NLB = AHEAD + 2                                         # rotating left batches: previous, current, AHEAD in flight
    lb = [slam.PyramidBatch((H, W), levels=levels, S=S, ctx=ctx) for _ in range(NLB)]
    built = [None] * NLB       # marker on the pyramid stream: "the build into this slot is complete"
    rb = slam.PyramidBatch((H, W), levels=levels, S=S, ctx=ctx)
    seq = frame_sequence(steps + warmup + 60 + S)
    rng = np.random.default_rng(1234)
    noise_pool = rng.normal(0, 0.5, (max(1 << 17, 4096 * S), 2))          # prior noise, drawn once (synthetic-input generation, not SLAM work)
    seq_a = np.asarray(seq); flows_a = np.asarray(flows, dtype=np.float64)
    lp = [t.data_ptr() for t in left_dev]; rp = [t.data_ptr() for t in right_dev]
    lptr = lambda i: [lp[f] for f in seq[i:i + S]]
    rptr = lambda i: [rp[f] for f in seq[i:i + S]]
    flow_at = lambda i: flows_a[seq_a[i:i + S]] - flows_a[seq_a[i - 1:i - 1 + S]]
    kp = np.zeros((0, 2)); is3d = np.zeros(0, dtype=bool); sid = np.zeros(0, dtype=np.int32)
    cur = 0

    nxt = [0]                  # next frame whose left build has not been enqueued yet

    def enqueue_build(frame):
        slot = frame % NLB
        lb[slot].update_(lptr(frame), sync=False, fast=fast, ctx=ctx_pyr)
        built[slot] = ctx_pyr.record(built[slot])

    def build_up_to(frame):
        while nxt[0] <= frame:
            enqueue_build(nxt[0]); nxt[0] += 1

    build_up_to(AHEAD)
    ctx_pyr.synchronize()
    kf_tail = os.environ.get("SLAM_BENCH_KF_TAIL", "0") != "0"     # measured 5 % slower when on
    state = dict(kp=kp, is3d=is3d, sid=sid, cur=cur, tracked=0)

    def step(i, pipelined=True):
        kf = (i - 1) % KF_EVERY == 0
        st_ = state
        prevb, curb = lb[(i - 1) % NLB], lb[i % NLB]
        if not pipelined:                                   # span pass: build this step's pyramids now, serially
            enqueue_build(i); nxt[0] = max(nxt[0], i + 1)
        if kf:
            rb.update_(rptr(i), sync=False, fast=fast, ctx=ctx_right, target_only=RIGHT_TARGET_ONLY)
        ctx.wait_event(built[i % NLB])                      # tracking needs the build of frame i only (i+1.. stay in flight)
        if pipelined:
            build_up_to(i + AHEAD)                          # overwrites the slot of a frame nothing reads any more
        kp, is3d, sid = st_["kp"], st_["is3d"], st_["sid"]
        if len(kp):
            o = (i * 7919) % (len(noise_pool) - len(kp))
            proj = flow_at(i).take(sid, axis=0); proj += kp; proj += noise_pool[o:o + len(kp)]
            # track + drop the keypoints whose tracking failed (map_manager.jl:523-560) in one call
            kp, is3d, sid, _ = slam.optical_flow_matching_batch_kept(prevb, curb, sid, kp, is3d, proj, params, ctx=ctx)
            st_["tracked"] += len(kp)
        if kf:
            if len(kp):
                keep = np.flatnonzero(rng.random(len(kp)) >= CULL_FRACTION)
                kp, is3d, sid = kp.take(keep, axis=0), is3d.take(keep), sid.take(keep)
            fresh, fsid = slam.detect_batch(extractor, curb, kp, sid, ctx=ctx)       # kp is kept grouped by stream
            if len(fresh):
                a = np.searchsorted(sid, np.arange(S + 1)); b = np.searchsorted(fsid, np.arange(S + 1))
                fresh = fresh.astype(np.float64)
                kp = np.concatenate([x for s_ in range(S) for x in (kp[a[s_]:a[s_ + 1]], fresh[b[s_]:b[s_ + 1]])])
                is3d = np.concatenate([x for s_ in range(S) for x in (is3d[a[s_]:a[s_ + 1]], np.zeros(b[s_ + 1] - b[s_], dtype=bool))])
                sid = np.concatenate([x for s_ in range(S) for x in (sid[a[s_]:a[s_ + 1]], fsid[b[s_]:b[s_ + 1]])])
            if pipelined and kf_tail:
                # the stereo match below is the tail of a key-frame step: the left / right builds are (nearly) done and the
                # match alone does not fill the GPU, so the build of frame i+AHEAD+1 starts now (its slot held frame i-1,
                # which the temporal match above was the last to read)
                build_up_to(i + AHEAD + 1)
            ctx.wait_for(ctx_right)
            proj = kp + np.array([0.0, -disparity])
            _, ok = slam.optical_flow_matching_batch(curb, rb, sid, kp, is3d, proj, params, ctx=ctx, status_only=True)
            is3d = is3d | ok
        st_["kp"], st_["is3d"], st_["sid"] = kp, is3d, sid
        if hook is not None and pipelined:
            hook()                                          # e.g. the pose seams of the step (synchronous, own context)

    def drain():
        ctx_pyr.synchronize(); ctx_right.synchronize(); ctx.synchronize(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    warm = max(warmup, 6)
    for i in range(1, 1 + warm):
        step(i)
    state["tracked"] = 0
    drain(); t0 = time.perf_counter()
    for i in range(1 + warm, 1 + warm + steps):
        step(i)
    drain(); dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    tracked = state["tracked"] / max(steps, 1) / S
    # per-kernel spans (serial launches, no pipelining) for the roofline of the batched launches
    for c in (ctx_pyr, ctx_right):
        c.prof_enable(True); c.prof_reset()
    base = 1 + warm + steps
    nprof = 20
    ctx_pyr.synchronize()
    for i in range(base, base + nprof):
        step(i, pipelined=False)
    drain()
    rows_ms, rows_n = [a + b for a, b in zip(ctx_pyr.prof_get("k_iir_rows"), ctx_right.prof_get("k_iir_rows"))]
    pyr_ms, pyr_n = [a + b for a, b in zip(ctx_pyr.prof_get("pyr_update"), ctx_right.prof_get("pyr_update"))]
    for c in (ctx_pyr, ctx_right):
        c.prof_enable(False)
    rb_bytes = S * iir_rows_bytes(H, W, levels) / (levels + 1)
    res = {"streams_per_gpu": S, "steps": steps, "value": world * S * steps / dt, "unit": "frames/sec", "seconds": dt,
           "ms_per_step_of_S_frames": dt / steps * 1e3, "tracked_kpts_per_frame": round(tracked, 1),
           "roofline": {"bound": "hbm", "kernel": "k_iir_seg<rows>" if (fast and S < 4) else "k_iir_rows (bit-exact kernels: batches of >= 4 images take them in mode 3 too)" if fast else "k_iir_rows", "achieved": rb_bytes / (rows_ms / max(rows_n, 1) * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rb_bytes / (rows_ms / max(rows_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "frac_of_achievable": rb_bytes / (rows_ms / max(rows_n, 1) * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS, "achievable_peak": HBM_ACHIEVABLE_GBS,
                        "avg_launch_us": rows_ms / max(rows_n, 1) * 1e3, "algorithmic_bytes_per_launch": rb_bytes, "traffic": None},
           "pyramid_batch_update_serial_us": pyr_ms / max(pyr_n, 1) * 1e3}
    pb = S * pyramid_bytes(H, W, levels)
    res["roofline"]["stage"] = {"name": f"pyramid update of {S} images (all kernels, serial launches)", "algorithmic_bytes": pb,
                                "avg_us": pyr_ms / max(pyr_n, 1) * 1e3, "achieved": pb / (pyr_ms / max(pyr_n, 1) * 1e-3) / 1e9,
                                "frac": pb / (pyr_ms / max(pyr_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "algorithmic = read the layer + write the 6 planes of every level once (SURVEY 8d); the separable filters and the "
                                        "two-dimensional recurrences need ~40 plane passes per level, which is what the kernels are bound by"}
    for c_ in (ctx, ctx_pyr, ctx_right):
        c_.close()
    return res


POSE_TOL_M = 0.1            # recovered camera translation vs the frames' image offsets (plane 30 m away), metres


WORKLOADS = {
    # name: shape (slam_jl_amd.synthetic.SHAPES), keypoints per frame, stereo, streams per GPU, camera (fx, fy, cx, cy), image step per frame
    "kitti05_1000": dict(ba_window="P5_free_20_const", shape="kitti05", kpts=1000, stereo=True, S=128, cam=None, step=(1.3, -2.1), n_frames=8,
                         what="BASELINE configs[1]: KITTI 05 stereo 370x1226, 1000 kpts/frame (the headline)"),
    "kitti00_2000": dict(ba_window="P20", shape="kitti00", kpts=2000, stereo=True, S=128, cam=None, step=(1.3, -2.1), n_frames=8,
                         what="BASELINE configs[2]: KITTI 00 stereo 376x1241 (example/kitty/main.jl:21-22), 2000 kpts/frame; one 20-KF local BA window per stream and key-frame: configs.kitti00_2000.with_ba (window alone: ba.windows.P20, ba.batch.P20)"),
    "euroc_mono": dict(ba_window="P50", shape="euroc", kpts=1000, stereo=False, S=128, cam=(458.654, 457.296, 367.215, 248.375), step=(2.6, -4.2), n_frames=8,
                       what="BASELINE configs[3]: monocular 480x640, PnP-tracking path (front_end.jl:132-219: five-point filter + P3P RANSAC + PnP "
                            "refinement every frame, no right image), new keypoints by triangulate_temporal!; one 50-KF BA window per stream and key-frame: configs.euroc_mono.with_ba (alone: ba.windows.P50, ba.batch.P50)"),
    "fhd_4000": dict(ba_window="P100", shape="fhd", kpts=4000, stereo=True, S=32, cam=(910.0, 910.0, 960.0, 540.0), step=(1.3, -2.1), n_frames=4,
                     what="BASELINE configs[4] on one GPU: 1080x1920 stereo (example/uni/main.jl:11-13), 4000 kpts/frame; one 100-KF BA window per stream and key-frame: configs.fhd_4000.with_ba (alone: ba.windows.P100, ba.batch.P100)"),
}


def make_workload(slam, syn, name, seed=0, streams=None):
    w = dict(WORKLOADS[name]); w["name"] = name
    H, W = syn.SHAPES[w["shape"]]
    camt = tuple(w["cam"] or syn.KITTI_CAM)
    params = slam.Params(stereo=w["stereo"], max_nb_keypoints=w["kpts"])
    cam = slam.Camera(*camt, height=H, width=W)
    w.update(H=H, W=W, camt=camt, params=params, extractor=slam.Extractor.from_params(params, cam), disparity=12.4, levels=params.pyramid_levels)
    if streams:
        w["S"] = streams
    w["left"], w["right"], w["flows"] = syn.stereo_stream(w["shape"], w["n_frames"], seed=seed, step=w["step"], disparity=w["disparity"])
    if not w["stereo"]:
        w["right"] = None
    return w



def run_lockstep_kpset(slam, torch, local_rank, wl, periods, warm_periods, world, dist, dev, ingest,
                       hook=None, seed=1234, pose=False, record=None, snapshot=None, diag=None, ba=None):
    """The headline loop: S streams in lock-step, keypoints resident in HBM (slam_kpset_*), no host list work
    between the calls of a frame; the host sees the S list lengths once per frame.  Timed: `periods` key-frame periods
    (KF_EVERY frames of every stream each, the first a key-frame) after `warm_periods` untimed ones.

    ingest: where a frame starts --
      "host_u8"  pinned host memory, 8-bit as the KITTI reader decodes them (example/kitty/kitty.jl:52-102): one H2D copy of
                 the S frames on the copy stream, converted on the device (slam_pyr_update_batch_u8_dev);
      "host_f64" pinned host memory as Matrix{Gray{Float64}} (what the Julia seam receives, SLAM.jl:250): 8x the bytes;
      "dev_f64"  already in HBM as Float64 (round-1 headline).
    Stream s plays the ping-pong sequence shifted by s frames, so the S frames of a step are a contiguous window of the
    periodic sequence: one copy per step.

    record (dict, optional): replay mode for the parity check -- runs record["frame_steps"] frames from the empty lists, no
      warm-up, and appends per frame {"i", "kf", "shift" (S, 2), "cull" (S, cap) uint8 or None} to record["steps"].
    snapshot (list of stream ids, optional): after the last frame, download those streams' keypoint lists and all planes of
      their current left pyramids into the result (the cpu_baseline leg compares them with the oracle)."""
    import ctypes as C
    S, H, W, params, extractor, camt, disparity = wl["S"], wl["H"], wl["W"], wl["params"], wl["extractor"], wl["camt"], wl["disparity"]
    left, right, flows, stereo = wl["left"], wl["right"], wl["flows"], wl["stereo"]
    fastpyr = bool(wl.get("tolerance"))                       # slam_pyr_update_batch mode 3: the tolerance-mode batch kernels (planes <= 1e-11 relative)
    # the tracking context's stream is in a scheduling class of its own (the low-priority one): a hardware queue that the branches
    # of the pyramid graph never land on.  With four default-class streams the runtime placed the graph's small-level branch on
    # the tracking stream's queue and every step's match sat behind it until the build was over (kernel trace).
    # Measured at S = 32, host_u8: default class 14.7k frames/s, high 16.2k, low 16.5k (the builds are the longer chain of a
    # step and are better left undisturbed).  SLAM_BENCH_TRACK_PRIO=0 restores the shared class for comparison.
    prio = int(os.environ.get("SLAM_BENCH_TRACK_PRIO", "-1"))
    pprio = int(os.environ.get("SLAM_BENCH_PYR_PRIO", "0"))
    # co-runner policies (measurement knobs): SLAM_BENCH_CU_SPLIT="T[:stride]" confines the tracking / detect context to T compute
    # units and the two pyramid contexts to the other 256 - T (contiguous CU numbers, or every k-th with ":stride");
    # SLAM_BENCH_SERIAL=1 puts everything on ONE stream (no overlap at all)
    split = os.environ.get("SLAM_BENCH_CU_SPLIT")
    serial = os.environ.get("SLAM_BENCH_SERIAL") is not None
    if split:
        T = int(split.split(":")[0]); stride = ":" in split
        ncu = 256
        if stride:
            k = max(1, round(ncu / max(T, 1))); tr = [1 if i % k == 0 else 0 for i in range(ncu)]
        else:
            tr = [1 if i >= ncu - T else 0 for i in range(ncu)]
        py = [1 - b for b in tr]
        ctx, ctx_pyr, ctx_right, ctx_copy = (slam.Context(local_rank, cu_mask=tr), slam.Context(local_rank, cu_mask=py), slam.Context(local_rank, cu_mask=py), copy_ctx(slam, local_rank))
    elif serial:
        ctx = leg_ctx(slam, local_rank); ctx_pyr = ctx_right = ctx; ctx_copy = copy_ctx(slam, local_rank)
    else:
        ctx, ctx_pyr, ctx_right, ctx_copy = (leg_ctx(slam, local_rank, prio), leg_ctx(slam, local_rank, pprio),
                                             leg_ctx(slam, local_rank, pprio), copy_ctx(slam, local_rank))
    levels = params.pyramid_levels
    AHEAD = max(1, int(os.environ.get("SLAM_BENCH_KP_AHEAD", "2")))   # builds enqueued ahead of the step being tracked (same-box A/B: 2 = +0.9 % over 1 -- the next graph is already queued when a build ends; 3 = -1.4 %)
    NLB = AHEAD + 3                                          # previous, current, AHEAD being built, one more being copied
    lb = [slam.PyramidBatch((H, W), levels=levels, S=S, ctx=ctx) for _ in range(NLB)]
    rb = slam.PyramidBatch((H, W), levels=levels, S=S, ctx=ctx) if stereo else None
    built = [None] * NLB
    copied = [None] * NLB; rcopied = [None]; rbuilt = [None]
    ncell = extractor.grid_resolution[0] * extractor.grid_resolution[1]
    cap = extractor.max_points + ncell + 8
    ks = slam.KeypointSet(S, cap, ctx=ctx)
    peek("run_lockstep_kpset: start")
    n_frames = len(left)
    period = 2 * n_frames - 2
    seq = frame_sequence_n(n_frames, period + S + 2)
    u8 = ingest == "host_u8"
    np_dtype, t_dtype, fbytes = (np.uint8, torch.uint8, H * W) if u8 else (np.float64, torch.float64, H * W * 8)
    conv = (lambda im: np.round(im * 255).astype(np.uint8)) if u8 else (lambda im: im)
    # the periodic frame sequence, contiguous (row-major (W, H) = Julia's column-major H x W)
    def seq_tensor(frames):
        a = np.stack([np.ascontiguousarray(conv(frames[seq[k]]).T) for k in range(period + S)])
        return torch.from_numpy(a)
    lseq = seq_tensor(left); rseq = seq_tensor(right) if stereo else None
    host = ingest != "dev_f64"
    if host:
        lseq = lseq.pin_memory()
        lstage = [torch.empty((S, W, H), dtype=t_dtype, device=dev) for _ in range(NLB)]
        if stereo:
            rseq = rseq.pin_memory()
            rstage = torch.empty((S, W, H), dtype=t_dtype, device=dev)
        st_copy = torch.cuda.ExternalStream(ctx_copy.stream, device=dev)         # H2D copies on their own stream, one step ahead of the builds
    else:
        lseq = lseq.to(dev); rseq = rseq.to(dev) if stereo else None
    torch.cuda.synchronize()
    flows_a = np.asarray(flows, dtype=np.float64); seq_a = np.asarray(seq)
    rng = np.random.default_rng(seed)
    st_main = torch.cuda.ExternalStream(ctx.stream, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(seed + local_rank)
    cull_u = torch.empty(S * cap, dtype=torch.float32, device=dev)      # allocated on torch's own stream: nothing is allocated inside the
    cull_flags = torch.zeros(S * cap, dtype=torch.bool, device=dev)     # library-stream contexts below (the caching allocator would keep using that stream)
    ev_pool = [(slam.Event(ctx_pyr, timed=True), slam.Event(ctx_pyr, timed=True)) for _ in range(48)]
    ev_used = []
    # hipEvents around the temporal match (k_kpset_match + the compaction behind it) on the TRACKING stream, timed region only
    lk_pool = [(slam.Event(ctx, timed=True), slam.Event(ctx, timed=True)) for _ in range(48)]
    lk_used = []                                             # (event pair, keypoints that entered the match)

    def ptrs(base_tensor):
        b = base_tensor.data_ptr()
        return [b + s * fbytes for s in range(S)]

    def enqueue_copy(frame):
        """the step's S left frames: pinned host -> staging slot, behind the last build that read the slot"""
        slot = frame % NLB
        if built[slot] is not None:
            ctx_copy.wait_event(built[slot])
        with torch.cuda.stream(st_copy):
            lstage[slot].copy_(lseq[frame % period:frame % period + S], non_blocking=True)
        copied[slot] = ctx_copy.record(copied[slot])

    def enqueue_build(frame, timed=False):
        slot = frame % NLB
        o = frame % period
        if host:
            ctx_pyr.wait_event(copied[slot])
            src = ptrs(lstage[slot])
        else:
            src = ptrs(lseq[o:o + S])
        if timed:
            pair = ev_pool[len(ev_used) % len(ev_pool)]
            ctx_pyr.record(pair[0])
        lb[slot].update_(src, sync=False, ctx=ctx_pyr, u8=u8, fast=fastpyr)
        if timed:
            ctx_pyr.record(pair[1]); ev_used.append(pair)
        built[slot] = ctx_pyr.record(built[slot])

    def enqueue_right_copy(frame):
        if rbuilt[0] is not None:
            ctx_copy.wait_event(rbuilt[0])
        with torch.cuda.stream(st_copy):
            rstage.copy_(rseq[frame % period:frame % period + S], non_blocking=True)
        rcopied[0] = ctx_copy.record(rcopied[0])

    def enqueue_right(frame):
        o = frame % period
        if host:
            ctx_right.wait_event(rcopied[0])
            src = ptrs(rstage)
        else:
            src = ptrs(rseq[o:o + S])
        rb.update_(src, sync=False, ctx=ctx_right, u8=u8, target_only=RIGHT_TARGET_ONLY, fast=fastpyr)
        rbuilt[0] = ctx_right.record(rbuilt[0])

    nxt = [0]; nxc = [0]
    def build_up_to(frame, timed=False):
        if host:
            while nxc[0] <= frame + 1:                       # copies run one frame ahead of the builds
                enqueue_copy(nxc[0]); nxc[0] += 1
        while nxt[0] <= frame:
            enqueue_build(nxt[0], timed); nxt[0] += 1

    Z_PLANE = 30.0
    baseline = disparity * Z_PLANE / camt[0]                 # a scene 30 m away: d = fx b / z
    T21 = np.eye(4); T21[0, 3] = -baseline
    Twc = np.eye(4)
    sp_stereo = slam.stream_params(S, cam=camt, shift_yx=np.tile([0.0, -disparity], (S, 1)))
    state = dict(n_bound=0, tracked=0, tracked_steps=0, timed=False, wait_s=0.0, booted=False)
    # pose = True: the full per-frame front-end of front_end.jl:60-113 on the tracked lists themselves -- the streams are a rigid
    # scene (a fronto-parallel plane 30 m away, cameras translating parallel to it), so the map points of the stereo
    # triangulation, the pose priors of the tracking, the five-point filter against the previous key-frame and P3P + PnP are all
    # consistent; the recovered camera translation is checked against the image offsets of the frames.
    pst = dict(Tcw=np.tile(np.eye(4), (S, 1, 1)), Tprev=np.tile(np.eye(4), (S, 1, 1)), Tkf=np.tile(np.eye(4), (S, 1, 1)), ref=None,
               accepted=0, asked=0, err_max=0.0, acc5=0, asked5=0, gated5=0, n_kf=0, kf_cw=np.tile(np.eye(4), (S, 8, 1, 1)))
    sp_cam = slam.stream_params(S, cam=camt)
    if host and stereo:
        enqueue_right_copy(1)                               # step 1 is a key-frame
    build_up_to(AHEAD)
    ctx_pyr.synchronize(); ctx_copy.synchronize()

    def step(i):
        kf = (i - 1) % KF_EVERY == 0
        prevb, curb = lb[(i - 1) % NLB], lb[i % NLB]
        if kf and stereo:
            enqueue_right(i)
        if host and stereo and i % KF_EVERY == 0:           # the next step is a key-frame: its right frames start travelling now
            enqueue_right_copy(i + 1)
        ctx.wait_event(built[i % NLB])                      # tracking needs the build of frame i only
        build_up_to(i + AHEAD, state["timed"])              # the next frame's copy + build overlap this step's tracking
        cnt = None
        rec = None
        if record is not None:
            rec = {"i": i, "kf": kf, "shift": None, "cull": None}; record["steps"].append(rec)
        if state["n_bound"] > 0 and pose:
            # klt_tracking! with the motion model's prediction (constant velocity on the translation), then the epipolar filter and
            # compute_pose!; the pose call is this step's device -> host copy
            Tpred = pst["Tcw"].copy(); Tpred[:, :3, 3] += pst["Tcw"][:, :3, 3] - pst["Tprev"][:, :3, 3]
            ks.flow_match(prevb, curb, params, slam.stream_params(S, Tcw=Tpred, cam=camt), prior=1, n_bound=state["n_bound"], ctx=ctx)
            Rc = np.tile(np.eye(4), (S, 1, 1)); Rc[:, :3, :3] = pst["Tkf"][:, :3, :3] @ np.transpose(Tpred[:, :3, :3], (0, 2, 1))
            r5 = ks.compute_pose_5pt(slam.stream_params(S, Tcw=Rc, cam=camt), min_parallax=5.0, max_repr_error=3.0, iters=128,
                                     seed=seed + 2 * i, ctx=ctx, fetch=(i % 4 == 0))      # enqueue-only on most steps: its effect is on the lists
            t_enq = time.perf_counter()
            poses, stp, _, cnt = ks.compute_pose(sp_cam, threshold=3.0, iters=256, seed=seed + 2 * i + 1, ctx=ctx)
            state["wait_s"] += time.perf_counter() - t_enq
            pst["Tprev"] = pst["Tcw"].copy()
            ok = stp.astype(bool)
            pst["Tcw"][ok] = poses[ok]
            if pst["ref"] is not None:
                off = flows_a[seq_a[(i % period) + np.arange(S)]] - pst["ref"]
                want = np.stack([off[:, 1] * Z_PLANE / camt[0], off[:, 0] * Z_PLANE / camt[1], np.zeros(S)], axis=1)
                pst["asked"] += S; pst["accepted"] += int(ok.sum())
                if r5 is not None:
                    st5 = np.asarray(r5[1]).astype(bool); par5 = np.asarray(r5[3])
                    pst["asked5"] += S; pst["acc5"] += int(st5.sum()); pst["gated5"] += int((~st5 & (par5 < 5.0)).sum())
                if ok.any():
                    pst["err_max"] = max(pst["err_max"], float(np.abs(pst["Tcw"][ok, :3, 3] - want[ok]).max()))
        elif state["n_bound"] > 0:
            # motion-model prior: the stream's image-plane shift, ~0.5 px off (project_world_to_image_distort of the map points
            # under the predicted pose; the synthetic streams are image-plane translations)
            shift = flows_a[seq_a[(i % period) + np.arange(S)]] - flows_a[seq_a[((i - 1) % period) + np.arange(S)]]
            shift = shift + rng.normal(0, 0.5, (S, 2))
            if rec is not None:
                rec["shift"] = shift.copy()
            sp = slam.stream_params(S, cam=camt, shift_yx=shift)
            lkp = None
            if state["timed"]:
                lkp = lk_pool[len(lk_used) % len(lk_pool)]; ctx.record(lkp[0])
            ks.flow_match(prevb, curb, params, sp, prior=2, n_bound=state["n_bound"], ctx=ctx)
            if lkp is not None:
                ctx.record(lkp[1]); lk_used.append((lkp, state["n_bound"]))
        if kf:
            # map culling between key-frames (outlier observations dropped by BA, failed triangulations): flags drawn in HBM
            with torch.cuda.stream(st_main):
                cull_u.uniform_(generator=gen)
                torch.lt(cull_u, CULL_FRACTION, out=cull_flags)                 # bool = one byte per slot, 1 = remove
            if rec is not None:
                ctx.synchronize()
                rec["cull"] = cull_flags.cpu().numpy().astype(np.uint8).reshape(S, cap)
            ks.remove(cull_flags.data_ptr(), ctx=ctx)
            ks.detect(extractor, curb, ctx=ctx)
            if pose and not stereo and not state["booted"]:
                # monocular initialisation taken as given (front_end.jl:243-332 + mapper.jl:185-262 run once at start-up): the first
                # key-frame's keypoints become map points on the scene plane; the loop measures the PnP-tracking steady state
                for s_ in range(S):
                    d_ = ks.download(s_, ctx=ctx)
                    yx_ = d_["yx"]
                    xyz_ = np.stack([(yx_[:, 1] - camt[2]) / camt[0] * Z_PLANE, (yx_[:, 0] - camt[3]) / camt[1] * Z_PLANE, np.full(len(yx_), Z_PLANE)], axis=1)
                    ks.upload(s_, yx_, np.ones(len(yx_), bool), xyz_, ids=d_["ids"], ctx=ctx)
                state["booted"] = True
            if pose:
                ks.keyframe(ctx=ctx)                        # the frame becomes the previous key-frame of its keypoints
                pst["Tkf"] = pst["Tcw"].copy()
                if pst["ref"] is None:                      # world frame = the first key-frame's camera
                    pst["ref"] = flows_a[seq_a[(i % period) + np.arange(S)]].copy()
            Twc_now = np.linalg.inv(pst["Tcw"]) if pose else Twc
            if stereo:
                ctx.wait_for(ctx_right)
                ks.stereo_match(curb, rb, params, sp_stereo, prior=2, ctx=ctx)
                ks.triangulate(camt, camt, T21, Twc_now, max_error=3.0, ctx=ctx)
            if pose:                                        # mapper.jl:86: what stereo left 2-D, against its first observing key-frame
                kfid = pst["n_kf"]; pst["kf_cw"][:, kfid % 8] = pst["Tcw"]; pst["n_kf"] += 1
                if kfid > 0:
                    ks.triangulate_temporal(sp_cam, pst["kf_cw"], Twc_now, kfid, max_error=3.0, ctx=ctx)
            cnt = None
        if kf and ba is not None:
            ba.submit()                                         # add_new_kf!(estimator, kf): this key-frame's local BA of every stream (task #3)
        if cnt is None:
            t_enq = time.perf_counter()
            cnt = ks.counts(ctx=ctx)                        # the one device -> host copy of the step (synchronises)
            state["wait_s"] += time.perf_counter() - t_enq  # host time spent waiting for the GPU (the rest of the step is enqueue work)
        tot = int(cnt.sum())
        if not kf and state["n_bound"] > 0:
            state["tracked"] += tot; state["tracked_steps"] += 1
        state["n_bound"] = tot
        if hook is not None:
            hook()
        if diag is not None:                                    # probes / tests: look at the lists and poses after every frame step
            diag(i, kf, pst, ks, ctx, flows_a[seq_a[(i % period) + np.arange(S)]])

    def drain():
        ctx_copy.synchronize(); ctx_pyr.synchronize(); ctx_right.synchronize(); ctx.synchronize(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if record is not None:
        warm, nsteps = 0, int(record["frame_steps"])
    else:
        warm, nsteps = max(warm_periods, 2) * KF_EVERY, periods * KF_EVERY
    for i in range(1, 1 + warm):
        step(i)
    state["tracked"] = 0; state["tracked_steps"] = 0
    if ba is not None:
        ba.reset_counters()
    drain(); state["timed"] = True; state["wait_s"] = 0.0; t0 = time.perf_counter()
    for i in range(1 + warm, 1 + warm + nsteps):
        step(i)
    if ba is not None:
        ba.join()                                               # the last key-frame's windows are part of the timed work
    drain(); dt = time.perf_counter() - t0
    dt_own = dt
    state["timed"] = False
    i_last = warm + nsteps
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    builds = [a.elapsed_ms(b) for a, b in ev_used[-len(ev_pool):]]      # left builds of the timed region (graph replays on the pyramid stream)
    lk_spans = [(a.elapsed_ms(b), n) for (a, b), n in lk_used[-len(lk_pool):]]
    try:
        free_b, total_b = torch.cuda.mem_get_info(dev)
        hbm_gb = (total_b - free_b) / 1e9                                # everything this process (and anyone else on the device) holds while the loop's buffers are alive
    except Exception:
        hbm_gb = None
    res = {"ingest": ingest, "streams_per_gpu": S, "hbm_in_use_gb": hbm_gb, "steps": periods, "frame_steps": nsteps, "value": world * S * nsteps / dt, "unit": "frames/sec", "seconds": dt,
           "value_this_rank": S * nsteps / dt_own, "seconds_this_rank": dt_own,
           "ms_per_step": dt / max(periods, 1) * 1e3, "ms_per_frame_of_S_streams": dt / nsteps * 1e3,
           "host_wait_ms_per_frame": state["wait_s"] / nsteps * 1e3,
           "local_ba": None if ba is None else ba.result(),
           "tracked_kpts_per_frame": round(state["tracked"] / max(state["tracked_steps"], 1) / S, 1),
           "pose": None if not pose else {"accepted_fraction": pst["accepted"] / max(pst["asked"], 1),
                                          "five_point_accepted_fraction": pst["acc5"] / max(pst["asked5"], 1),
                                          "five_point_rejected_by_parallax_gate_fraction": pst["gated5"] / max(pst["asked5"], 1),
                                          "five_point_note": "compute_pose_5pt! returns nothing while the average parallax against the previous key-frame is below 5 px "
                                                             "(front_end.jl:290): the first frames after each key-frame; every remaining call is accepted when the two fractions add up to 1",
                                          "max_translation_error_m": pst["err_max"], "plane_depth_m": Z_PLANE,
                                          # the loop's own check: every compute_pose! accepted and the recovered translation within POSE_TOL_M of the
                                          # frames' offsets (a 30 m scene, translations of a few metres); bench.py turns a failure into leg_error
                                          "pose_ok": bool(pst["asked"] > 0 and pst["accepted"] == pst["asked"] and pst["err_max"] < POSE_TOL_M),
                                          "pose_tol_m": POSE_TOL_M},
           "lk_match": None if not lk_spans else {"mean_ms": float(np.mean([m for m, _ in lk_spans])), "points_per_launch": float(np.mean([n for _, n in lk_spans])),
                                                  "n": len(lk_spans), "what": "hipEvents around slam_kpset_flow_match (k_kpset_match + compaction) on the tracking stream, timed region"},
           "pyramid_build_ms": {"mean": float(np.mean(builds)) if builds else None, "min": float(np.min(builds)) if builds else None,
                                "n": len(builds), "what": "hipEvents around each left-batch build (one hipGraph replay, u8 ingest fused) on the pyramid "
                                                          "stream inside the timed region, tracking running beside it"}}
    if snapshot is not None:
        snap = {}
        curb = lb[i_last % NLB]
        for s_ in snapshot:
            snap[s_] = {"frame_id": int(seq[(i_last % period) + s_]), "list": ks.download(s_, ctx=ctx),
                        "planes": {(nm, l): curb.pyramids[s_].plane(nm, l, ctx=ctx) for l in range(levels + 1)
                                   for nm in ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")}}
        res["snapshot"] = snap
        res["seq"] = seq; res["period"] = period; res["cap"] = cap
    peek("run_lockstep_kpset: after the loop")
    for c in (ctx, ctx_pyr, ctx_right, ctx_copy):
        c.synchronize()
    torch.cuda.synchronize()
    peek("run_lockstep_kpset: after the synchronisation")
    ks.close()
    peek("run_lockstep_kpset: after ks.close")
    for e2 in ev_pool + lk_pool:
        e2[0].close(); e2[1].close()
    for m in built + copied + rcopied + rbuilt:
        if m is not None:
            m.close()
    peek("run_lockstep_kpset: after closing events / markers")
    for b_ in lb + ([rb] if rb is not None else []):
        for p_ in b_.pyramids:
            p_.close()
    peek("run_lockstep_kpset: after destroying the pyramids")
    del lseq, rseq, cull_u, cull_flags
    if host:
        del lstage, st_copy
    del st_main
    retire_torch_host_events(torch)
    peek("run_lockstep_kpset: after freeing the torch buffers")
    for c_ in ({id(c): c for c in (ctx, ctx_pyr, ctx_right)}).values():      # (the copy context lives as long as the process: copy_ctx)
        c_.close()
    return res


# the local-BA windows a stream owes per key-frame: the reference's own cap (5 free key-frames + their constant observers,
# estimator.jl:327-331) and the window sizes BASELINE's configs name (every point seen by 10 consecutive key-frames, the first pose constant)
BA_WINDOW_SHAPES = {
    "P5_free_20_const": (lambda syn, z: syn.ba_scene(P=25, M=800, seed=100 + z, n_const=20),
                         "5 free + 20 constant key-frames, 800 points, 8000 observations (estimator.jl:327-331), 5 + 10 LM iterations", 8),
    "P20": (lambda syn, z: syn.ba_scene(P=20, M=4000, seed=300 + z), "20 key-frames (19 free), 4000 points, 40 000 observations, 5 + 10 LM iterations", 8),
    "P50": (lambda syn, z: syn.ba_scene(P=50, M=10000, seed=500 + z), "50 key-frames (49 free), 10 000 points, 100 000 observations, 5 + 10 LM iterations", 4),
    "P100": (lambda syn, z: syn.ba_scene(P=100, M=40000, seed=700 + z), "100 key-frames (99 free), 40 000 points, 400 000 observations, 5 + 10 LM iterations", 2),
}


class BAWorker:
    """The reference's estimator task (#3, estimator.jl:78-99) for S lock-stepped streams: every key-frame step hands the S windows of the
    streams (reference-shaped: 5 free + 20 constant key-frames, estimator.jl:327-331) to slam_local_ba_batch on a context and a host thread
    of their own; the front-end loop goes on.  A new hand-over first waits for the previous solve (the estimator takes key-frames in order),
    so a solve slower than a key-frame period shows in the loop's frame rate.  The windows are synthetic (the array contract of
    _get_ba_parameters, estimator.jl:143-266, filled by synthetic.ba_scene): map bookkeeping stays on the host in the reference."""

    def __init__(self, slam, syn, local_rank, S, prio=0, window="P5_free_20_const"):
        import threading
        self.threading = threading
        self.ctx = leg_ctx(slam, local_rank, prio)
        mk, self.window_text, nbase = BA_WINDOW_SHAPES[window]
        base = [mk(syn, z) for z in range(nbase)]
        caches = [slam.LocalBACache(base[z % nbase]["theta0"].copy(), base[z % nbase]["theta_const"], base[z % nbase]["pixels_yx"], base[z % nbase]["pose_ids"],
                                    base[z % nbase]["point_ids"]) for z in range(S)]
        self.batch = slam.BABatch(caches, base[0]["cam"])
        self.window = window
        self.S = S; self.th = None; self.calls = 0; self.wall = 0.0; self.wait = 0.0; self.err = None
        self.batch.solve(ctx=self.ctx, reset=True)                  # warm-up: scratch, pinned block, function attributes

    def _run(self):
        try:
            t0 = time.perf_counter()
            self.batch.solve(ctx=self.ctx, reset=True)
            self.wall += time.perf_counter() - t0; self.calls += 1
        except Exception as ex:                                     # noqa: BLE001
            self.err = ex

    def join(self):
        if self.th is not None:
            t0 = time.perf_counter(); self.th.join(); self.wait += time.perf_counter() - t0; self.th = None
        if self.err is not None:
            raise self.err

    def submit(self):
        self.join()
        self.th = self.threading.Thread(target=self._run); self.th.start()

    def reset_counters(self):
        self.join(); self.calls = 0; self.wall = 0.0; self.wait = 0.0

    def result(self):
        self.join()
        ok = bool((self.batch.status == 0).all())
        return {"windows_per_call": self.S, "calls": self.calls, "mean_call_ms": self.wall / max(self.calls, 1) * 1e3,
                "front_end_waited_ms_per_call": self.wait / max(self.calls, 1) * 1e3, "all_windows_ok": ok,
                "device_ms_per_call": float(self.batch.stats[0, 6]), "window_name": self.window, "window": self.window_text}

    def close(self):
        self.join(); self.ctx.close()


def leg_ctx(slam, local_rank, priority=0):
    """a context (HIP stream, scratch, pinned block) for one leg; the leg closes it"""
    return slam.Context(local_rank, priority=priority) if priority else slam.Context(local_rank)


_COPY_CTX = {}


def copy_ctx(slam, local_rank):
    """The context whose stream PyTorch copies pinned host frames on (torch.cuda.ExternalStream) lives as long as the process.  PyTorch's
    pinned-memory allocator records an event on that stream for every non-blocking copy and queries those events on LATER allocations;
    once in ~17 full runs a later pin_memory() failed with hipErrorCapturedEvent ("event last recorded in a capturing stream") after the
    stream had been destroyed with its leg -- although nothing in the process captures a stream any more: the runtime's
    answer to querying an event whose stream is gone.  Keeping this ONE stream (and retiring the allocator's events before the other
    contexts of a leg are closed, retire_torch_host_events) removes the situation; the library's own contexts need no such care."""
    if local_rank not in _COPY_CTX:
        _COPY_CTX[local_rank] = slam.Context(local_rank)
    return _COPY_CTX[local_rank]


def retire_torch_host_events(torch):
    """after the leg's pinned tensors are gone and the device is idle: one small pinned allocation makes PyTorch's host allocator walk its
    pending events (all complete by now) while the streams they were recorded on still exist"""
    torch.cuda.synchronize()
    torch.empty(64, dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()


_HIP = None


def peek(label):
    """SLAM_BENCH_PEEK=1: report the calling thread's pending HIP error (hipPeekAtLastError of the runtime torch and the library share)
    -- to find the call that leaves hipErrorStreamCaptureUnsupported behind for a later torch call to trip over"""
    global _HIP
    if os.environ.get("SLAM_BENCH_PEEK") is None:
        return
    import ctypes, sys as _s
    if _HIP is None:
        import torch as _t
        libdir = os.path.join(os.path.dirname(_t.__file__), "lib")
        _HIP = ctypes.CDLL(os.path.join(libdir, "libamdhip64.so"))
    e = _HIP.hipPeekAtLastError()
    if e != 0:
        print(f"[peek] pending HIP error {e} at {label}", file=_s.stderr, flush=True)


def kernel_spans(slam, torch, local_rank, wl, dev):
    fast = bool(wl.get("tolerance"))
    """Per-kernel device time of the batched build with serial launches (hipEvent spans cannot look inside the graph),
    and the graph replay alone on the GPU."""
    S, H, W, params, left = wl["S"], wl["H"], wl["W"], wl["params"], wl["left"]
    ctx = leg_ctx(slam, local_rank)
    pb = slam.PyramidBatch((H, W), levels=params.pyramid_levels, S=S, ctx=ctx)
    seq = frame_sequence_n(len(left), S + 2)
    t = torch.from_numpy(np.stack([np.ascontiguousarray(np.round(left[seq[k]] * 255).astype(np.uint8).T) for k in range(S)])).to(dev)
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() + s * H * W for s in range(S)]
    peek("kernel_spans: before the first build")
    pb.update_(ptrs, sync=True, ctx=ctx, u8=True, fast=fast)
    peek("kernel_spans: after the first build (capture)")
    ea, eb = slam.Event(ctx, timed=True), slam.Event(ctx, timed=True)
    ctx.record(ea)
    for _ in range(20):                                      # the stage alone on the GPU: graph replays back to back
        pb.update_(ptrs, sync=False, ctx=ctx, u8=True, fast=fast)
    ctx.record(eb)
    isolated_us = ea.elapsed_ms(eb) / 20 * 1e3
    peek("kernel_spans: after the replays")
    ea.close(); eb.close()
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(20):
        pb.update_(ptrs, sync=False, ctx=ctx, u8=True, fast=fast)
    ctx.synchronize()
    rows_ms, rows_n = ctx.prof_get("k_iir_rows"); pyr_ms, pyr_n = ctx.prof_get("pyr_update")
    ctx.prof_enable(False)
    peek("kernel_spans: after the profiled builds")
    for p_ in pb.pyramids:
        p_.close()
    peek("kernel_spans: after destroying the pyramids")
    ctx.close()
    return rows_ms / max(rows_n, 1) * 1e3, pyr_ms / max(pyr_n, 1) * 1e3, isolated_us


