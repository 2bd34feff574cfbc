"""The reference's per-frame call protocol on arrays (Stream) over three back ends: the single-image GPU seams (GpuBackend), the same with a
key-frame period per batched build (GpuPeriodBackend), and the CPU oracle (CpuBackend -- handed in by bench.py's cpu_baseline leg;
nothing here imports the oracle)."""
import numpy as np

from .common import KF_EVERY, RIGHT_TARGET_ONLY, CULL_FRACTION


class Stream:
    """The reference's per-frame call protocol on arrays: preprocess! (pyramid
    swap + update!, front_end.jl:454-470), optical_flow_matching! for tracked
    keypoints (map_manager.jl:451-564), and at key-frames extract_keypoints!
    (map_manager.jl:98-113) + right pyramid update! + stereo matching
    (mapper.jl:51-66).  `be` supplies the five seams (GPU product or CPU oracle)."""

    def __init__(self, be, flows, disparity, seed=0):
        self.be, self.flows, self.disparity = be, flows, disparity
        self.kp = np.zeros((0, 2)); self.is3d = np.zeros(0, dtype=bool)
        self.rng = np.random.default_rng(seed)
        self.t = 0
        self.n_tracked = 0
        # the prior's noise is INPUT (a stand-in for the motion model's error): drawn once, before any timed region, and read in turn --
        # drawing 2 x n normals per frame inside the loop cost the single-stream loop ~15 us of numpy per 290-us frame
        self.noise = self.rng.normal(0, 0.5, (1 << 17, 2)); self.noise_at = 0
        self.dflow = {}

    def step(self, f_prev, f_cur, upcoming=()):
        be = self.be
        kf = self.t % KF_EVERY == 0
        be.kf_next = (self.t + 1) % KF_EVERY == 0            # the workload's key-frame cadence is fixed: a backend may request the next key-frame's right pyramid early
        be.begin_frame(f_cur, upcoming, kf)
        if len(self.kp):
            flow = self.dflow.get((f_prev, f_cur))
            if flow is None:
                flow = self.dflow[(f_prev, f_cur)] = np.array(self.flows[f_cur]) - np.array(self.flows[f_prev])
            n = len(self.kp)
            if self.noise_at + n > len(self.noise): self.noise_at = 0
            proj = self.kp + flow + self.noise[self.noise_at:self.noise_at + n]       # motion-model prior, ~0.5 px off
            self.noise_at += n
            new, st = be.match(False, self.kp, self.is3d, proj)
            self.kp, self.is3d = new[st], self.is3d[st]
            self.n_tracked += int(st.sum())
        if kf:
            # map culling between key-frames (outlier observations dropped by BA, estimator.jl:283-292;
            # failed triangulations, mapper.jl:142-263): the synthetic scene never loses tracks by itself
            if len(self.kp):
                keep = self.rng.random(len(self.kp)) >= CULL_FRACTION
                self.kp, self.is3d = self.kp[keep], self.is3d[keep]
            fresh = be.detect(self.kp)
            if len(fresh):
                self.kp = np.concatenate([self.kp, fresh.astype(np.float64)])
                self.is3d = np.concatenate([self.is3d, np.zeros(len(fresh), dtype=bool)])
            proj = self.kp + np.array([0.0, -self.disparity])
            _, st = be.match(True, self.kp, self.is3d, proj)
            self.is3d = self.is3d | st                                            # stereo-matched -> triangulated
        self.t += 1


class GpuBackend:
    """One stereo stream through the single-image entry points (latency view).  Contexts (HIP streams) mirror the reference's tasks:
    `ctx` tracks / detects (front-end), `ctx_right` builds the right pyramid of a key-frame (mapper, mapper.jl:52), and the left
    pyramids are built AHEAD of the tracking on `ahead` build contexts in turn: the pyramid of frame t+k does not depend on the
    tracking result of frame t, a recorded sequence (example/kitty/main.jl reads its frames from disk) has the next frames at hand,
    and a single-image build leaves most of the chip idle -- so `ahead` builds are in flight while frame t is tracked
    (ahead = 1: the next frame only, the configuration of rounds 1-2).  ahead + 2 pyramids rotate so that a build never overwrites
    planes still being read; markers (slam_event) order the tracking behind the one build it needs."""

    def __init__(self, slam, ctx, ctx_pyr, ctx_right, H, W, left_dev, right_dev, params, extractor, pipelined=True, fast=False, ahead=1, extra_build_ctx=()):
        self.slam, self.ctx, self.ctx_right, self.params, self.e = slam, ctx, ctx_right, params, extractor
        self.build_ctx = [ctx_pyr] + list(extra_build_ctx)[:max(ahead - 1, 0)]
        self.left, self.right, self.pipelined, self.fast = left_dev, right_dev, pipelined, fast
        self.ahead = max(1, ahead)
        self.npyr = self.ahead + 2
        self.pyr = [slam.LKPyramid(shape=(H, W), levels=params.pyramid_levels, ctx=ctx) for _ in range(self.npyr)]
        self.rpyrs = [slam.LKPyramid(shape=(H, W), levels=params.pyramid_levels, ctx=ctx) for _ in range(2)]
        self.rpyr = self.rpyrs[0]
        self.rheld = [None, None]                # frame number whose right image each right pyramid holds / is being built with
        self.rbuilt = [None, None]               # marker: that build is complete
        self.kf_next = False
        self.built = [None] * self.npyr          # marker: "the build into this slot is complete"
        self.holds = [None] * self.npyr          # (frame number, image id) the slot holds or is being built with
        self.i = 0                 # frame number of the current frame; slot = i % npyr

    @property
    def cur(self):
        return self.pyr[self.i % self.npyr]

    @property
    def prev(self):
        return self.pyr[(self.i - 1) % self.npyr]

    def _build(self, t, f, sync=False):
        slot = t % self.npyr
        c = self.build_ctx[t % len(self.build_ctx)]
        self.slam.update_(self.pyr[slot], None, device_ptr=self.left[f].data_ptr(), sync=sync, ctx=c, fast=self.fast, chain=self.ahead > 1)
        self.built[slot] = c.record(self.built[slot])
        self.holds[slot] = (t, f)

    def _build_right(self, t, f):
        # with several build streams the right build goes FIRST onto the stream whose left build the tracking has just waited for (its
        # queue is empty; the next left build of that stream is enqueued behind it): one more stream would be one more than the GPU has
        # hardware queues, and the right build would sit behind whatever it aliased with
        c = self.build_ctx[self.i % len(self.build_ctx)] if self.ahead > 1 else self.ctx_right
        self.slam.update_(self.rpyrs[t % 2], None, device_ptr=self.right[f].data_ptr(), sync=False, ctx=c, fast=self.fast,
                          target_only=RIGHT_TARGET_ONLY, chain=self.ahead > 1)
        self.rbuilt[t % 2] = c.record(self.rbuilt[t % 2])
        self.rheld[t % 2] = t

    def prime(self, f):
        self._build(self.i, f, sync=True)

    def begin_frame(self, f_cur, upcoming, kf):
        self.i += 1                                       # copy!(prev, cur) as a handle rotation (pyramid.jl:28)
        if self.holds[self.i % self.npyr] != (self.i, f_cur) or not self.pipelined:
            self._build(self.i, f_cur)
        if kf:                                            # right image of a key-frame, on its own stream (mapper task, mapper.jl:52)
            self.rpyr = self.rpyrs[self.i % 2]
            if self.rheld[self.i % 2] != self.i:
                self._build_right(self.i, f_cur)
        self.ctx.wait_event(self.built[self.i % self.npyr])        # tracking below needs the build of THIS frame only
        if self.pipelined and self.ahead > 1 and self.kf_next and len(upcoming):
            # the next frame is a key-frame (fixed cadence of the workload): its right pyramid is requested now, so that the stereo
            # match does not sit behind a 400-700 us build (the reference's mapper thread builds it beside the front-end, mapper.jl:52)
            self._build_right(self.i + 1, list(upcoming)[0])
        if self.pipelined:
            for k, f in enumerate(list(upcoming)[:self.ahead], 1):
                if self.holds[(self.i + k) % self.npyr] != (self.i + k, f):
                    self._build(self.i + k, f)

    def match(self, stereo, kp, is3d, proj):
        a, b = (self.cur, self.rpyr) if stereo else (self.prev, self.cur)
        if stereo:
            self.ctx.wait_event(self.rbuilt[self.i % 2])
        return self.slam.optical_flow_matching(a, b, kp, is3d, proj, self.params, ctx=self.ctx)

    def detect(self, cur):
        return self.slam.detect(self.e, self.cur, cur, ctx=self.ctx)

    def drain(self):
        for c in self.build_ctx:
            c.synchronize()
        self.ctx_right.synchronize(); self.ctx.synchronize()

    def close(self):
        for m in self.built + self.rbuilt:
            if m is not None:
                m.close()
        for p_ in self.pyr + self.rpyrs:
            p_.close()


class GpuPeriodBackend:
    """One stereo stream whose next KEY-FRAME PERIOD is built in one batched launch set: a recorded sequence has its next frames at hand,
    and the library builds S images per launch with the bit-exact kernels (slam_pyr_update_batch_dev) -- the chain-bound single-image
    kernels of five independent builds in flight leave the GPU mostly idle, one batch of the period's five left frames + the key-frame's
    right frame costs little more than one image.  Tracking, detection and stereo matching go through the single-image entry points on the
    batch's member pyramids, every call synchronous, exactly as in GpuBackend; period k + 1 is requested on the build context when
    period k's first frame is reached (three batches rotate: the last member of period k - 1 is still `prev` then)."""

    def __init__(self, slam, ctx, ctx_build, H, W, left_dev, right_dev, params, extractor, period):
        self.slam, self.ctx, self.cb, self.params, self.e = slam, ctx, ctx_build, params, extractor
        self.left, self.right, self.B = left_dev, right_dev, period
        self.batches = [slam.PyramidBatch((H, W), levels=params.pyramid_levels, S=period + 1, ctx=ctx_build) for _ in range(3)]
        self.built = [None, None, None]              # marker: the build into that batch is complete
        self.first = slam.LKPyramid(shape=(H, W), levels=params.pyramid_levels, ctx=ctx)
        self.i = 0; self.requested = -1
        self.cur = self.prev = self.rpyr = None
        self.kf_next = False

    def prime(self, f):
        self.slam.update_(self.first, None, device_ptr=self.left[f].data_ptr(), sync=True, ctx=self.ctx)
        self.cur = self.first

    def _request(self, k, frames):
        """period k: its B left frames + the right frame of its first (key-)frame, one batched build"""
        b = self.batches[k % 3]
        ptrs = [self.left[f].data_ptr() for f in frames] + [self.right[frames[0]].data_ptr()]
        b.update_(ptrs, sigma=self.params.pyramid_sigma, sync=False, ctx=self.cb)
        self.built[k % 3] = self.cb.record(self.built[k % 3])
        self.requested = k

    def begin_frame(self, f_cur, upcoming, kf):
        self.i += 1
        k, m = divmod(self.i - 1, self.B)
        up = list(upcoming)
        if m == 0:
            assert kf, "the period of the batches is the key-frame cadence"
            if self.requested < k:                                            # the very first period
                self._request(k, [f_cur] + up[:self.B - 1])
            self.ctx.wait_event(self.built[k % 3])
            if len(up) >= 2 * self.B - 1:
                self._request(k + 1, up[self.B - 1:2 * self.B - 1])
        b = self.batches[k % 3]
        self.prev, self.cur = self.cur, b.pyramids[m]
        if kf:
            self.rpyr = b.pyramids[self.B]

    def match(self, stereo, kp, is3d, proj):
        a, b = (self.cur, self.rpyr) if stereo else (self.prev, self.cur)
        return self.slam.optical_flow_matching(a, b, kp, is3d, proj, self.params, ctx=self.ctx)

    def detect(self, cur):
        return self.slam.detect(self.e, self.cur, cur, ctx=self.ctx)

    def drain(self):
        self.cb.synchronize(); self.ctx.synchronize()

    def close(self):
        for m in self.built:
            if m is not None:
                m.close()
        self.first.close()
        for b in self.batches:
            for p_ in b.pyramids:
                p_.close()


class CpuBackend:
    """The CPU oracle on the same protocol (cpu_baseline leg only)."""

    def __init__(self, orc, left, right, params, extractor, threads):
        self.orc, self.left, self.right, self.params, self.e, self.threads = orc, left, right, params, extractor, threads
        self.prev = self.cur = self.rpyr = None
        self.img = None

    def prime(self, f):
        self.cur = self.orc.pyr_build(self.left[f], self.params.pyramid_levels, self.params.pyramid_sigma, 1)

    def begin_frame(self, f_cur, upcoming, kf):
        self.prev = self.cur
        self.img = self.left[f_cur]
        self.cur = self.orc.pyr_build(self.img, self.params.pyramid_levels, self.params.pyramid_sigma, 1)
        if kf:
            self.rpyr = self.orc.pyr_build(self.right[f_cur], self.params.pyramid_levels, self.params.pyramid_sigma, 1)

    def _fb(self, a, b, pts, disp, levels):
        return self.orc.fb_tracking(a, b, pts, disp, 30, self.params.window_size, levels, 1e-4, 1e-2,
                                    self.params.max_ktl_distance, sum_order=0, threads=self.threads)

    def match(self, stereo, kp, is3d, proj):
        a, b = (self.cur, self.rpyr) if stereo else (self.prev, self.cur)
        n = len(kp); new = kp.copy(); status = np.zeros(n, dtype=bool)
        ids3 = np.where(is3d)[0]; ids2 = list(np.where(~is3d)[0])
        if len(ids3):
            nk, st = self._fb(a, b, kp[ids3], 0.5 * (proj[ids3] - kp[ids3]), 1)
            new[ids3[st]] = nk[st]; status[ids3[st]] = True; ids2 += list(ids3[~st])
        if len(ids2):
            ids2 = np.asarray(ids2)
            nk, st = self._fb(a, b, kp[ids2], None, self.params.pyramid_levels)
            new[ids2[st]] = nk[st]; status[ids2[st]] = True
        return new, status

    def detect(self, cur):
        return self.orc.detect(self.img, cur, max_points=self.e.max_points, radius=self.e.radius, cell_size=self.e.cell_size)


