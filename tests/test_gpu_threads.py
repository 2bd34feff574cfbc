"""GPU: the THREADING contract of the seams.  The reference calls the hot path from three tasks at once:
task 1, the front-end (SLAM.jl:166,187-230: update! of the left pyramid, optical_flow_matching!, detect); task 2, the mapper
(mapper.jl:26,37-66: update! of the right pyramid, stereo matching from a deepcopy of the left key-frame pyramid, SLAM.jl:218);
task 3, the estimator (estimator.jl:79-99: bundle_adjustment!).  Here three OS threads (ctypes releases the GIL during a call)
drive libslamhip concurrently, one slam_ctx each, for 200 iterations, and every result must be bit-equal to the same calls
issued one after the other.  One thread also creates and destroys a context (its stream, scratch, pinned block) per iteration
while the others are mid-call -- the pattern that tripped the stream-capture state of rounds 1-3; the pyramid build
graph is now constructed node by node (no capture)."""
import queue
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H, W = 370, 1226
N_ITERS = 200


def _frames(syn, n=8):
    L, R, flows = syn.stereo_stream((H, W), n, seed=5, disparity=12.4)
    u8 = lambda im: np.asfortranarray(np.round(im * 255).astype(np.uint8))
    return [u8(x) for x in L], [u8(x) for x in R], flows


class _Work:
    """the three tasks' bodies on given contexts; every call returns plain numpy results"""

    def __init__(self, slam, syn, orc):
        self.slam = slam
        self.L, self.R, self.flows = _frames(syn)
        self.params = slam.Params(stereo=True, max_nb_keypoints=300)
        cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
        self.ex = slam.Extractor.from_params(self.params, cam)
        img0 = np.asfortranarray(self.L[0].astype(np.float64) / 255.0)
        self.kp = orc.detect(img0, np.zeros((0, 2)), max_points=300).astype(np.float64)
        self.is3d = (np.arange(len(self.kp)) % 4 != 0)
        self.scene = syn.ba_scene(P=25, M=300, seed=5, n_const=20)

    def frame(self, i):
        return i % len(self.L)

    def front_end(self, ctx, prev, cur, i):
        """task 1: update!(cur, frame i) -> optical_flow_matching!(prev -> cur) -> detect on the new frame"""
        s = self.slam
        s.update_(cur, self.L[self.frame(i)])
        shift = np.array(self.flows[self.frame(i)]) - np.array(self.flows[self.frame(i - 1)])
        new, st = s.optical_flow_matching(prev, cur, self.kp, self.is3d, self.kp + shift, self.params, ctx=ctx)
        det = s.detect(self.ex, cur, new[st], ctx=ctx)
        return new, st, det

    def mapper(self, ctx, left_clone, right, i):
        """task 2: update!(right, frame i) (only ever matched INTO) -> stereo matching from the key-frame's left pyramid"""
        s = self.slam
        s.update_(right, self.R[self.frame(i)], target_only=True)
        proj = self.kp + np.array([0.0, -12.4])
        new, st = s.optical_flow_matching(left_clone, right, self.kp, np.zeros(len(self.kp), bool), proj, self.params, ctx=ctx)
        return new, st

    def estimator(self, ctx):
        """task 3: bundle_adjustment! on the reference's window shape (5 free + 20 constant key-frames)"""
        s, sc = self.slam, self.scene
        cache = s.LocalBACache(sc["theta0"].copy(), sc["theta_const"], sc["pixels_yx"], sc["pose_ids"], sc["point_ids"])
        s.bundle_adjustment_(cache, sc["cam"], ctx=ctx)
        return cache.theta.copy(), cache.outliers.copy()


def _same(a, b):
    return all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))


def test_three_tasks_concurrently_equal_the_serial_run(slam, syn, orc):
    w = _Work(slam, syn, orc)
    # ---- the serial run: one context, the same calls in program order ----
    c0 = slam.Context(0)
    prev = slam.LKPyramid(shape=(H, W), levels=3, ctx=c0); cur = slam.LKPyramid(shape=(H, W), levels=3, ctx=c0)
    right = slam.LKPyramid(shape=(H, W), levels=3, ctx=c0)
    slam.update_(prev, w.L[0])
    ref_fe, ref_map = [], []
    n_ref = 2 * len(w.L)                                   # the frame sequence is periodic: iteration i repeats iteration i - period
    for i in range(1, 1 + n_ref):
        ref_fe.append(w.front_end(c0, prev, cur, i))
        clone = slam.deepcopy(cur)
        ref_map.append(w.mapper(c0, clone, right, i))
        clone.close()
        prev, cur = cur, prev
    ref_ba = w.estimator(c0)
    for p in (prev, cur, right):
        p.close()
    c0.close()
    period = len(w.L)
    ref_of = lambda lst, i: lst[(i - 1) % period + (period if i > period else 0)]      # iterations > period are in steady state (prev from the same cycle)

    # ---- three threads ----
    errors, handoff = [], queue.Queue(maxsize=4)
    bad = {"fe": 0, "map": 0, "ba": 0}

    def task1():
        try:
            c = slam.Context(0)
            prev = slam.LKPyramid(shape=(H, W), levels=3, ctx=c); cur = slam.LKPyramid(shape=(H, W), levels=3, ctx=c)
            slam.update_(prev, w.L[0])
            for i in range(1, 1 + N_ITERS):
                r = w.front_end(c, prev, cur, i)
                if not _same(r, ref_of(ref_fe, i)):
                    bad["fe"] += 1
                handoff.put((i, slam.deepcopy(cur)))           # SLAM.jl:218: the key-frame's pyramid goes to the mapper as a deep copy
                prev, cur = cur, prev
            handoff.put(None)
            prev.close(); cur.close(); c.close()
        except Exception as ex:                                # noqa: BLE001 -- reported by the main thread
            errors.append(("task1", repr(ex))); handoff.put(None)

    def task2():
        try:
            c = slam.Context(0)
            right = slam.LKPyramid(shape=(H, W), levels=3, ctx=c)
            while True:
                item = handoff.get()
                if item is None:
                    break
                i, clone = item
                r = w.mapper(c, clone, right, i)
                if not _same(r, ref_of(ref_map, i)):
                    bad["map"] += 1
                clone.close()
            right.close(); c.close()
        except Exception as ex:                                # noqa: BLE001
            errors.append(("task2", repr(ex)))
            while handoff.get() is not None:                   # keep task 1 from blocking on a full queue
                pass

    def task3():
        try:
            for _ in range(N_ITERS):
                c = slam.Context(0)                            # created and destroyed inside the loop, the others mid-call
                r = w.estimator(c)
                c.close()
                if not _same(r, ref_ba):
                    bad["ba"] += 1
        except Exception as ex:                                # noqa: BLE001
            errors.append(("task3", repr(ex)))

    threads = [threading.Thread(target=f) for f in (task1, task2, task3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a task did not finish"
    assert not errors, errors
    assert bad == {"fe": 0, "map": 0, "ba": 0}, bad


def test_two_threads_construct_their_build_graphs_at_once(slam, syn, orc):
    """the build graph of a pyramid batch is constructed lazily by its first update (explicit kernel nodes): two threads, each with
    its own context and batch, do that at the same moment, five times over -- planes equal the oracle's"""
    import torch
    S = 8
    L, _, _ = _frames(syn, 2)
    dev = torch.from_numpy(np.stack([np.ascontiguousarray(L[s % 2].T) for s in range(S)])).cuda()
    torch.cuda.synchronize()
    ptrs = [dev.data_ptr() + s * H * W for s in range(S)]
    ref = orc.pyr_build(np.asfortranarray(L[1].astype(np.float64) / 255.0), 3, 1.0, 1)
    for _ in range(5):
        cs = [slam.Context(0), slam.Context(0)]
        pbs = [slam.PyramidBatch((H, W), levels=3, S=S, ctx=c) for c in cs]
        go = threading.Barrier(2)
        errs = []

        def run(k):
            try:
                go.wait()
                pbs[k].update_(ptrs, u8=True, ctx=cs[k])
                pbs[k].update_(ptrs, u8=True, ctx=cs[k], fast=True)      # a second graph (tolerance mode) of the same batch
                pbs[k].update_(ptrs, u8=True, ctx=cs[k])
            except Exception as ex:                            # noqa: BLE001
                errs.append(repr(ex))

        ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not errs, errs
        for k in range(2):
            for name in ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx"):
                assert np.array_equal(pbs[k].pyramids[S - 1].plane(name, 1, ctx=cs[k]), ref.plane(name, 1)), (k, name)
            for p_ in pbs[k].pyramids:
                p_.close()
        for c in cs:
            c.close()
