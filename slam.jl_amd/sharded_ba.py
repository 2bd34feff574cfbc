"""Point-sharded local bundle adjustment over N GPUs.

One process per GPU.  Map points (with all their observations) are partitioned
across ranks; every rank holds all poses.  Per LM iteration each rank builds its
contribution to the reduced camera system on its GPU (libslamhip
`slam_ba_build`), ONE `all_reduce(SUM)` over RCCL/xGMI combines
[S (6P x 6P) ; rhs ; diag(Jp'Jp) ; ssr], every rank solves the small dense
system redundantly and back-substitutes its own points (`slam_ba_solve`), and a
4-double all-reduce combines the trial / predicted costs so that all ranks take
the same accept/reject decision.  The LM outer loop is LeastSquaresOptim's
(reference: src/bundle_adjustment.jl:35-54 and SURVEY Appendix A.8).

The reference has no distributed path; its BA window is <= 5 free key-frames
(src/estimator.jl:327-331), for which this path never pays off -- it exists for
the 50/100-KF windows BASELINE.json names and is gated by `worth_sharding`.
"""
import ctypes as C

import numpy as np

from . import _lib as L

LM_MAX_DELTA, LM_MIN_DELTA = 1e16, 1e-16
LM_MIN_STEP_QUALITY = 1e-3
LM_DELTA0 = 10.0
LM_XTOL = LM_FTOL = 1e-8


def worth_sharding(P, O, world_size):
    """Gate (north_star: 'only when the window is large enough to amortise it'), calibrated on round-2 measurements of one
    MI355X (profiles/r02*_ba_kernel_stats.csv, P = 50, O = 1e5): per LM iteration the observation-proportional part
    (k_linearize, k_points, k_obs_factors, k_blocks, k_backsub, k_trial) takes ~1.45 us per 1000 observations and is what
    sharding divides by the world size; the reduced-system solve (~3 us per key-frame, banded) is repeated on every rank;
    the exchange costs one all-reduce of 8 (6P)^2 bytes (ring: 2 (N-1)/N of the bytes over one ~150 GB/s xGMI link,
    ~25 us floor), one 32-byte all-gather (~15 us) and ~10 extra launches (~20 us) per iteration."""
    if world_size <= 1:
        return False
    build_us = 1.45e-3 * O
    saved_us = build_us * (1.0 - 1.0 / world_size)
    allreduce_us = 25.0 + 2.0 * (world_size - 1) / world_size * 8.0 * (6 * P) ** 2 / 150e3
    overhead_us = allreduce_us + 15.0 + 20.0
    return saved_us > 1.5 * overhead_us          # margin: the model is per iteration, set-up (pair lists, upload) is not sharded away


def sharding_model(P, O, world_size):
    """the numbers behind worth_sharding, per LM iteration (us): what N ranks save on the observation-proportional part and what the exchange costs"""
    build_us = 1.45e-3 * O
    saved_us = build_us * (1.0 - 1.0 / max(world_size, 1))
    allreduce_us = 25.0 + 2.0 * (world_size - 1) / max(world_size, 1) * 8.0 * (6 * P) ** 2 / 150e3
    return {"build_us_one_gpu": build_us, "saved_us": saved_us, "exchange_us": allreduce_us + 15.0 + 20.0, "margin": 1.5}


def predicted_crossover(world_size, obs_per_keyframe=2000):
    """smallest window (key-frames P, with obs_per_keyframe observations each: 2000 in the 20 / 50-KF windows of bench.py, 4000 in the 100-KF one) for which worth_sharding says yes at this
    world size, or None below 512 key-frames -- the model's prediction, to be held against the first measured multi-GPU curve"""
    for P in range(2, 513):
        if worth_sharding(P, obs_per_keyframe * P, world_size):
            return P
    return None


def partition_points(point_ids, M, world_size):
    """Contiguous point ranges balanced by observation count.
    Returns a list of (m_lo, m_hi) 0-based half-open ranges, one per rank."""
    cnt = np.bincount(np.asarray(point_ids, dtype=np.int64) - 1, minlength=M).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(cnt)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        bounds.append(int(np.searchsorted(cum, target, side="left")))
    bounds.append(M)
    bounds = np.maximum.accumulate(np.minimum(bounds, M))
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world_size)]


class HipShard:
    """This rank's shard on its GPU (libslamhip slam_ba_*); buffers are torch CUDA tensors."""

    def __init__(self, cam, P, theta_local, theta_const, pixels, pose_ids, point_ids_local, ctx=None):
        import torch
        self.torch = torch
        self.ctx = ctx or L.default_context(torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.P = P
        self.M = (len(theta_local) - 6 * P) // 3
        self.O = len(pose_ids)
        th = np.ascontiguousarray(theta_local, dtype=np.float64)
        tc = np.ascontiguousarray(theta_const, dtype=np.uint8)
        px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
        pi = np.ascontiguousarray(pose_ids, dtype=np.int64)
        li = np.ascontiguousarray(point_ids_local, dtype=np.int64)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.slam_ba_create(self.ctx.h, cam[0], cam[1], cam[2], cam[3], P, self.M, self.O,
                                                   L.ptr(th), L.ptr(tc, L.u8p), L.ptr(px), L.ptr(pi, L.i64p), L.ptr(li, L.i64p),
                                                   C.byref(h)))
        self.h = h
        dev = torch.device("cuda", self.ctx.device)
        self.red = torch.zeros(self.ctx.lib.slam_ba_reduce_len(P), dtype=torch.float64, device=dev)
        self.trial = torch.zeros(4, dtype=torch.float64, device=dev)
        # torch fills these on ITS stream; libslamhip runs on its own non-blocking stream
        torch.cuda.synchronize(dev)

    def build(self, ignore_outliers, inv_delta):
        self.ctx.check(self.ctx.lib.slam_ba_build(self.ctx.h, self.h, int(ignore_outliers), float(inv_delta), C.c_void_p(self.red.data_ptr())))
        return self.red

    def solve(self, red, inv_delta):
        self.ctx.check(self.ctx.lib.slam_ba_solve(self.ctx.h, self.h, C.c_void_p(red.data_ptr()), float(inv_delta), C.c_void_p(self.trial.data_ptr())))
        return self.trial

    def commit(self, accept):
        self.ctx.check(self.ctx.lib.slam_ba_commit(self.ctx.h, self.h, int(accept)))

    def flag_outliers(self, repr_eps, depth_eps=1e-6):
        n = C.c_int(0)
        self.ctx.check(self.ctx.lib.slam_ba_flag_outliers(self.ctx.h, self.h, float(repr_eps), float(depth_eps), C.byref(n)))
        return n.value

    # ---- device-paced pass (slam_ba_lm_* + slam_comm_*): no host synchronisation inside a pass ----
    def make_comm(self, group=None):
        """RCCL communicator over the ranks of `group` through the C ABI (slam_comm_*); the 128-byte id travels over
        torch.distributed's object broadcast (any backend) -- a Julia host would use its own channel."""
        import torch.distributed as dist
        lib = self.ctx.lib
        ws = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank(group) if ws > 1 else 0
        idbuf = C.create_string_buffer(128)
        # librccl prints a version banner on stdout when it initialises: keep the caller's stdout clean
        import os
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            return self._make_comm(group, lib, ws, rank, idbuf, dist)
        finally:
            sys.stdout.flush()
            C.CDLL(None).fflush(None)                        # the banner sits in C stdio's buffer: flush it while fd 1 still points at stderr
            os.dup2(saved, 1); os.close(saved)

    def _make_comm(self, group, lib, ws, rank, idbuf, dist):
        if rank == 0:
            self.ctx.check(lib.slam_comm_unique_id(idbuf))
        if ws > 1:
            box = [bytes(idbuf.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            idbuf = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        self.ctx.check(lib.slam_comm_create(self.ctx.h, ws, rank, idbuf, C.byref(h)))
        self.comm, self.ws = h, ws
        self.gathered = self.torch.zeros(4 * ws, dtype=self.torch.float64, device=self.red.device)
        self.torch.cuda.synchronize(self.red.device)
        return self

    def halfband(self):
        """block half-bandwidth of THIS shard's contribution to the reduced camera system"""
        return int(self.ctx.lib.slam_ba_halfband(self.h))

    def set_halfband(self, hb):
        """the all-reduced system is as wide as the widest shard's: every rank solves with the maximum over the ranks"""
        self.ctx.check(self.ctx.lib.slam_ba_set_halfband(self.h, int(hb)))

    def measure_collectives(self, reps=20):
        """device time (us) of one all-reduce of the reduce buffer and of one all-gather of the trial costs on this shard's
        communicator, hipEvents on the library stream (the constants of worth_sharding, measured where they run)"""
        from ._lib import Event
        lib, c = self.ctx.lib, self.ctx
        scratch = self.torch.zeros_like(self.red); self.torch.cuda.synchronize(self.red.device)
        red, n_red = C.c_void_p(scratch.data_ptr()), scratch.numel()
        trial, gath = C.c_void_p(self.trial.data_ptr()), C.c_void_p(self.gathered.data_ptr())
        out = {}
        for name, call in (("allreduce", lambda: lib.slam_comm_allreduce_sum(c.h, self.comm, red, n_red)),
                           ("allgather", lambda: lib.slam_comm_allgather(c.h, self.comm, trial, gath, 4))):
            c.check(call())
            a, b = Event(c, timed=True), Event(c, timed=True)
            c.record(a)
            for _ in range(reps):
                c.check(call())
            c.record(b)
            out[name] = a.elapsed_ms(b) / reps * 1e3
            a.close(); b.close()
        out["allreduce_bytes"] = n_red * 8
        return out

    def lm_pass(self, ignore, iters, first_pass):
        """One LM pass (bundle_adjustment.jl:35-54): returns (ssr at the start, ssr at the end, iterations)."""
        lib, c, red = self.ctx.lib, self.ctx, C.c_void_p(self.red.data_ptr())
        n_red = self.red.numel()
        c.check(lib.slam_ba_lm_begin(c.h, self.h, int(ignore), red))
        c.check(lib.slam_comm_allreduce_sum(c.h, self.comm, red, n_red))
        c.check(lib.slam_ba_lm_start(c.h, self.h, red, int(first_pass)))
        st0 = np.zeros(8)
        if iters == 0:
            c.check(lib.slam_ba_lm_state(c.h, self.h, L.ptr(st0)))
            return st0[0], st0[0], 0
        trial, gath = C.c_void_p(self.trial.data_ptr()), C.c_void_p(self.gathered.data_ptr())
        for it in range(1, iters + 1):
            if it > 1:
                c.check(lib.slam_ba_lm_build(c.h, self.h, int(ignore), red))
                c.check(lib.slam_comm_allreduce_sum(c.h, self.comm, red, n_red))
            c.check(lib.slam_ba_lm_solve(c.h, self.h, red, int(ignore), trial))
            c.check(lib.slam_comm_allgather(c.h, self.comm, trial, gath, 4))
            c.check(lib.slam_ba_lm_step(c.h, self.h, gath, self.ws, it))
        st = np.zeros(8)
        c.check(lib.slam_ba_lm_state(c.h, self.h, L.ptr(st)))
        if st[4] != 0.0:
            raise L.SlamHipError("sharded BA: reduced camera system not positive definite")
        return None, st[0], int(st[1])

    def download(self):
        theta = np.zeros(6 * self.P + 3 * self.M)
        outl = np.zeros(max(self.O, 1), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.slam_ba_download(self.ctx.h, self.h, L.ptr(theta), L.ptr(outl, L.u8p)))
        return theta, outl[:self.O].astype(bool)

    def close(self):
        if getattr(self, "comm", None):
            self.ctx.synchronize()
            self.ctx.lib.slam_comm_destroy(self.comm)
            self.comm = None
        if getattr(self, "h", None):
            self.ctx.lib.slam_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _all_reduce(t, op, group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        backend = str(dist.get_backend(group)).lower()          # "nccl", "gloo", or a composite such as "cuda:nccl,cpu:gloo" (default init)
        if t.is_cuda and "nccl" not in backend:
            # a host-only backend (gloo: two processes sharing one GPU, where RCCL refuses a second rank per device): staged through
            # the host.  ONLY the host-paced callers reach this branch (slam_ba_build / slam_ba_solve synchronise the library's
            # non-blocking stream before returning, so .cpu() on torch's stream sees the finished buffer); the device-paced driver
            # uses slam_comm_* and never comes here
            h = t.detach().cpu()
            dist.all_reduce(h, op=op, group=group)
            t.copy_(h)
            import torch
            torch.cuda.current_stream(t.device).synchronize()
            return t
        dist.all_reduce(t, op=op, group=group)
        if t.is_cuda:
            # RCCL runs on torch's streams, libslamhip on its own non-blocking stream: the
            # reduced buffer must be complete before slam_ba_solve is enqueued
            import torch
            torch.cuda.current_stream(t.device).synchronize()
    return t


def sharded_bundle_adjustment(cam, theta, theta_const, pixels, pose_ids, point_ids, iterations=10, repr_eps=5.0,
                              iters_fast=5, group=None, shard_factory=HipShard, timings=None, host_paced=None):
    """bundle_adjustment! over all ranks of `group`.  Every rank passes the FULL
    problem (as the single-GPU seam would receive it) and gets the FULL result:
    (theta (6P+3M), outliers (O,), stats).  `shard_factory` builds this rank's
    compute shard (HipShard in the product; the tests inject an oracle-backed
    one to check the partition + collective logic on CPU/gloo).  With the product's HipShard the pass runs device-paced
    (slam_ba_lm_* + RCCL through slam_comm_*: no host synchronisation inside a pass); `host_paced=True` forces the
    host-driven loop over torch.distributed that the CPU shards use."""
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if ws > 1 else 0
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    tc = np.ascontiguousarray(theta_const, dtype=np.uint8)
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    pi = np.ascontiguousarray(pose_ids, dtype=np.int64)
    li = np.ascontiguousarray(point_ids, dtype=np.int64)
    P, O = len(tc), len(pi)
    M = (len(theta) - 6 * P) // 3
    n = 6 * P
    parts = partition_points(li, M, ws)
    m_lo, m_hi = parts[rank]
    sel = np.where((li - 1 >= m_lo) & (li - 1 < m_hi))[0]
    theta_local = np.concatenate([theta[:n], theta[n + 3 * m_lo:n + 3 * m_hi]])
    SUM, MAX = dist.ReduceOp.SUM, dist.ReduceOp.MAX

    def agree(err):
        """every rank learns whether ANY rank failed (a rank that raised alone would leave its peers in the next collective for ever)"""
        if ws > 1:
            flag = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64,
                                device=torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=MAX, group=group)
            if err is None and float(flag[0]) != 0.0:
                raise L.SlamHipError("sharded BA: another rank failed to set up its shard")
        if err is not None:
            raise err

    shard, err = None, None
    try:
        shard = shard_factory(cam, P, theta_local, tc, px[sel], pi[sel], li[sel] - m_lo)
    except Exception as ex:                                  # noqa: BLE001 -- re-raised on every rank by agree()
        err = ex
    agree(err)
    device_paced = hasattr(shard, "lm_pass") and host_paced is not True
    if device_paced:
        try:
            shard.make_comm(group)                           # RCCL through the C ABI on the library's own stream
        except Exception as ex:                              # noqa: BLE001
            err = ex
        agree(err)
    if hasattr(shard, "halfband"):
        # the all-reduced system has the widest shard's band: agreed BEFORE the first solve of either path (a shard solving the summed
        # system with its own narrower band would ignore blocks its peers contributed)
        hbt = torch.tensor([float(shard.halfband())], dtype=torch.float64, device=shard.red.device if hasattr(shard, "red") else "cpu")
        _all_reduce(hbt, MAX, group)
        shard.set_halfband(int(hbt[0]))

    def run_pass(ignore, iters):
        delta, decrease = LM_DELTA0, 2.0
        red = _all_reduce(shard.build(ignore, 1.0 / delta), SUM, group)
        ssr = float(red[n * n + 2 * n])
        ssr0 = ssr
        it, converged = 0, False
        while not converged and it < iters:
            it += 1
            if it > 1:
                red = _all_reduce(shard.build(ignore, 1.0 / delta), SUM, group)
            tr = shard.solve(red, 1.0 / delta)
            mx = tr[2:3].clone()
            _all_reduce(tr[0:2], SUM, group)
            _all_reduce(mx, MAX, group)
            trial, pred, maxdx = float(tr[0]), float(tr[1]), float(mx[0])
            if float(tr[3]) != 0.0:
                raise L.SlamHipError("sharded BA: reduced camera system not positive definite")
            rho = (trial - ssr) / (pred - ssr)
            if rho > LM_MIN_STEP_QUALITY:
                x_conv = maxdx <= LM_XTOL
                f_conv = abs(ssr - trial) / (abs(ssr) + LM_FTOL) <= LM_FTOL
                ssr = trial
                u = 2.0 * rho - 1.0
                delta = min(delta / max(1.0 / 3.0, 1.0 - u * u * u), LM_MAX_DELTA)
                decrease = 2.0
                shard.commit(1)
                converged = x_conv or f_conv
            else:
                delta = max(delta / decrease, LM_MIN_DELTA)
                decrease *= 2.0
                shard.commit(0)
                converged = maxdx <= LM_XTOL
        return ssr0, ssr, it

    import time as _time
    t_lm = 0.0
    if device_paced:
        # product path: LM decisions on the device, no host synchronisation inside a pass
        t0 = _time.perf_counter()
        _, ssr1, it1 = shard.lm_pass(0, iters_fast, True)
        t_lm += _time.perf_counter() - t0
        st0 = np.zeros(8); shard.ctx.check(shard.ctx.lib.slam_ba_lm_state(shard.ctx.h, shard.h, L.ptr(st0)))
        ssr_init = st0[5]
    else:
        ssr_init, ssr1, it1 = run_pass(0, iters_fast)
    n_out = torch.tensor([shard.flag_outliers(repr_eps)], dtype=torch.float64, device=shard.red.device if hasattr(shard, "red") else "cpu")
    _all_reduce(n_out, SUM, group)
    if device_paced:
        t0 = _time.perf_counter()
        _, ssr2, it2 = shard.lm_pass(1, iterations, False)
        t_lm += _time.perf_counter() - t0
    else:
        _, ssr2, it2 = run_pass(1, iterations)
    th_loc, ol_loc = shard.download()
    # gather the full result on every rank (poses are identical everywhere)
    theta_out = theta.copy()
    outl = np.zeros(O, dtype=bool)
    if ws > 1:
        objs = [None] * ws
        dist.all_gather_object(objs, (m_lo, m_hi, th_loc, sel, ol_loc), group=group)
    else:
        objs = [(m_lo, m_hi, th_loc, sel, ol_loc)]
    for lo, hi, th, s, ol in objs:
        theta_out[:n] = th[:n]
        theta_out[n + 3 * lo:n + 3 * hi] = th[n:]
        outl[s] = ol
    stats = dict(ssr_init=ssr_init, ssr_pass1=ssr1, ssr_final=ssr2, iters_pass1=it1, iters_pass2=it2,
                 n_outliers=int(n_out[0]), world_size=ws, points_local=m_hi - m_lo, obs_local=len(sel),
                 lm_wall_ms=t_lm * 1e3 if device_paced else None,
                 collectives_us=shard.measure_collectives() if (device_paced and timings is not None) else None)
    if hasattr(shard, "close"):
        shard.close()
    return theta_out, outl, stats
