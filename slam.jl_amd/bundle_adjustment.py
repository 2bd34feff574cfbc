"""LocalBACache / bundle_adjustment! / pnp_bundle_adjustment mirrors (reference:
src/estimator.jl:16-40, src/bundle_adjustment.jl)."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L


@dataclass
class LocalBACache:
    """The array part of LocalBACache (estimator.jl:16-40).  θ = [6P ; 3M],
    pixels (O, 2) as (y, x), ids 1-based."""
    theta: np.ndarray
    theta_const: np.ndarray
    pixels: np.ndarray
    poses_ids: np.ndarray
    points_ids: np.ndarray
    outliers: np.ndarray = None
    stats: dict = field(default_factory=dict)

    @property
    def n_poses(self):
        return len(self.theta_const)

    @property
    def n_points(self):
        return (len(self.theta) - 6 * self.n_poses) // 3


def bundle_adjustment_(cache, camera, iterations=10, repr_eps=5.0, iters_fast=5, ctx=None):
    """bundle_adjustment!(cache, camera; iterations, repr_ϵ) -- mutates cache.theta / cache.outliers."""
    ctx = ctx or L.default_context()
    fx, fy, cx, cy = camera.intrinsics if hasattr(camera, "intrinsics") else camera
    theta = np.ascontiguousarray(cache.theta, dtype=np.float64)
    tc = np.ascontiguousarray(cache.theta_const, dtype=np.uint8)
    px = np.ascontiguousarray(cache.pixels, dtype=np.float64).reshape(-1, 2)
    pi = np.ascontiguousarray(cache.poses_ids, dtype=np.int64)
    li = np.ascontiguousarray(cache.points_ids, dtype=np.int64)
    P, O = len(tc), len(pi)
    M = (len(theta) - 6 * P) // 3
    outl = np.zeros(max(O, 1), dtype=np.uint8)
    st = np.zeros(8)
    ctx.check(ctx.lib.slam_local_ba(ctx.h, fx, fy, cx, cy, P, M, O, L.ptr(theta), L.ptr(tc, L.u8p), L.ptr(px),
                                    L.ptr(pi, L.i64p), L.ptr(li, L.i64p), L.ptr(outl, L.u8p),
                                    int(iters_fast), int(iterations), float(repr_eps), L.ptr(st)))
    cache.theta = theta
    cache.outliers = outl[:O].astype(bool)
    cache.stats = dict(ssr_init=st[0], ssr_pass1=st[1], ssr_final=st[2], iters_pass1=int(st[3]), iters_pass2=int(st[4]),
                       n_outliers=int(st[5]), device_ms=st[6])
    return cache


class BABatch:
    """The arrays of S LocalBACaches stored back to back as `slam_local_ba_batch` takes them (packed once; `theta0` is kept so that the
    same windows can be solved again: bench.py).  cameras: one (fx, fy, cx, cy) / Camera or one per cache."""

    def __init__(self, caches, cameras):
        S = self.S = len(caches)
        one = hasattr(cameras, "intrinsics") or np.ndim(cameras) == 1
        self.cams = np.asarray([(c.intrinsics if hasattr(c, "intrinsics") else c) for c in ([cameras] * S if one else cameras)], dtype=np.float64).reshape(S, 4)
        th = [np.ascontiguousarray(c.theta, dtype=np.float64) for c in caches]
        tc = [np.ascontiguousarray(c.theta_const, dtype=np.uint8) for c in caches]
        px = [np.ascontiguousarray(c.pixels, dtype=np.float64).reshape(-1, 2) for c in caches]
        pi = [np.ascontiguousarray(c.poses_ids, dtype=np.int64) for c in caches]
        li = [np.ascontiguousarray(c.points_ids, dtype=np.int64) for c in caches]
        self.Pn = np.array([len(t) for t in tc], dtype=np.int32); self.On = np.array([len(t) for t in pi], dtype=np.int32)
        self.Mn = np.array([(len(t) - 6 * p) // 3 for t, p in zip(th, self.Pn)], dtype=np.int32)
        cat = lambda xs, dt, shape=(0,): np.ascontiguousarray(np.concatenate(xs)) if sum(len(x) for x in xs) else np.zeros(shape, dtype=dt)
        self.theta0 = cat(th, np.float64); self.theta = self.theta0.copy()
        self.tc = cat(tc, np.uint8); self.px = cat(px, np.float64, (0, 2)); self.pi = cat(pi, np.int64); self.li = cat(li, np.int64)
        self.outl = np.zeros(max(int(self.On.sum()), 1), dtype=np.uint8)
        self.stats = np.zeros((S, 8)); self.status = np.zeros(S, dtype=np.int32)
        self.th_off = np.concatenate([[0], np.cumsum(6 * self.Pn.astype(np.int64) + 3 * self.Mn)])
        self.ob_off = np.concatenate([[0], np.cumsum(self.On.astype(np.int64))])

    def solve(self, iterations=10, repr_eps=5.0, iters_fast=5, ctx=None, reset=False):
        """one `slam_local_ba_batch` call on the packed arrays (theta in place; reset: start again from theta0)"""
        ctx = ctx or L.default_context()
        if reset:
            self.theta[:] = self.theta0
        ctx.check(ctx.lib.slam_local_ba_batch(ctx.h, self.S, L.ptr(self.cams), L.ptr(self.Pn, L.i32p), L.ptr(self.Mn, L.i32p), L.ptr(self.On, L.i32p),
                                              L.ptr(self.theta), L.ptr(self.tc, L.u8p), L.ptr(self.px), L.ptr(self.pi, L.i64p), L.ptr(self.li, L.i64p),
                                              L.ptr(self.outl, L.u8p), int(iters_fast), int(iterations), float(repr_eps), L.ptr(self.stats),
                                              L.ptr(self.status, L.i32p)))
        return self.status

    def begin(self, iterations=10, repr_eps=5.0, iters_fast=5, ctx=None, reset=False):
        """`slam_local_ba_batch_begin`: the call proceeds on a thread of the library; `end()` waits for it.  Until then the context is the
        job's and the packed arrays must not be touched."""
        ctx = ctx or L.default_context()
        if reset:
            self.theta[:] = self.theta0
        ctx.check(ctx.lib.slam_local_ba_batch_begin(ctx.h, self.S, L.ptr(self.cams), L.ptr(self.Pn, L.i32p), L.ptr(self.Mn, L.i32p), L.ptr(self.On, L.i32p),
                                                    L.ptr(self.theta), L.ptr(self.tc, L.u8p), L.ptr(self.px), L.ptr(self.pi, L.i64p), L.ptr(self.li, L.i64p),
                                                    L.ptr(self.outl, L.u8p), int(iters_fast), int(iterations), float(repr_eps), L.ptr(self.stats),
                                                    L.ptr(self.status, L.i32p)))
        self._job_ctx = ctx

    def end(self):
        ctx = self._job_ctx; self._job_ctx = None
        ctx.check(ctx.lib.slam_local_ba_batch_end(ctx.h))
        return self.status

    def window(self, z):
        """(theta, outliers, stats dict) of window z after solve()"""
        st = self.stats[z]
        return (self.theta[self.th_off[z]:self.th_off[z + 1]].copy(), self.outl[self.ob_off[z]:self.ob_off[z + 1]].astype(bool),
                dict(ssr_init=st[0], ssr_pass1=st[1], ssr_final=st[2], iters_pass1=int(st[3]), iters_pass2=int(st[4]), n_outliers=int(st[5]),
                     device_ms=st[6], status=int(self.status[z])))


def bundle_adjustment_batch_(caches, cameras, iterations=10, repr_eps=5.0, iters_fast=5, ctx=None):
    """bundle_adjustment! for a list of LocalBACaches in one set of launches (`slam_local_ba_batch`): the estimator tasks of S
    lock-stepped SlamManagers (estimator.jl:78-99).  cameras: one (fx, fy, cx, cy) / Camera or one per cache.  Mutates every cache
    (theta, outliers, stats); returns the per-window status codes (0 = ok, SLAM_ERR_NUMERIC = that window left unchanged)."""
    b = BABatch(caches, cameras)
    status = b.solve(iterations=iterations, repr_eps=repr_eps, iters_fast=iters_fast, ctx=ctx)
    for z, c in enumerate(caches):
        c.theta, c.outliers, c.stats = b.window(z)
    return status.copy()


def ba_plan_order(cache):
    """-> (order, half_bandwidth, reordered): the pose order bundle_adjustment_ solves `cache` in (order[k] = the cache's 0-based pose
    at the solver's position k) and the block half-bandwidth of the reduced camera system in it -- `slam_ba_plan_order`, host work only.
    A window with loop closures (map_manager.jl:300-449) is not banded in key-frame order; folded, it is."""
    lib = L.load()
    tc = np.ascontiguousarray(cache.theta_const, dtype=np.uint8)
    pi = np.ascontiguousarray(cache.poses_ids, dtype=np.int64)
    li = np.ascontiguousarray(cache.points_ids, dtype=np.int64)
    P, O = len(tc), len(pi)
    M = (len(cache.theta) - 6 * P) // 3
    order = np.zeros(P, dtype=np.int32); hb = C.c_int(0)
    rc = lib.slam_ba_plan_order(P, M, O, L.ptr(tc, L.u8p), L.ptr(pi, L.i64p), L.ptr(li, L.i64p), L.ptr(order, L.i32p), C.byref(hb))
    if rc < 0:
        raise L.SlamHipError(f"slam_ba_plan_order: bad arguments ({rc})")
    return order, int(hb.value), bool(rc)


def pnp_bundle_adjustment(camera, pose, pixels, points, iterations=10, depth_eps=1e-6, repr_eps=5.0, iters_fast=5, ctx=None):
    """-> (new_pose 4x4, initial_error, final_error, outliers, n_outliers) -- bundle_adjustment.jl:113-171"""
    ctx = ctx or L.default_context()
    fx, fy, cx, cy = camera.intrinsics if hasattr(camera, "intrinsics") else camera
    pose = np.asfortranarray(pose, dtype=np.float64)
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    n = len(px)
    out = np.zeros((4, 4), order="F")
    e0, e1, no = C.c_double(), C.c_double(), C.c_int()
    outl = np.zeros(max(n, 1), dtype=np.uint8)
    ctx.check(ctx.lib.slam_pnp_ba(ctx.h, fx, fy, cx, cy, L.ptr(pose), L.ptr(px), L.ptr(pts), n, int(iters_fast), int(iterations),
                                  float(depth_eps), float(repr_eps), L.ptr(out), C.byref(e0), C.byref(e1),
                                  L.ptr(outl, L.u8p), C.byref(no)))
    return np.array(out), e0.value, e1.value, outl[:n].astype(bool), no.value


def pnp_bundle_adjustment_batch(cameras, poses, pixels, points, iterations=10, depth_eps=1e-6, repr_eps=5.0, iters_fast=5, ctx=None):
    """S single-pose refinements in one launch (lists of per-stream arrays; cameras: one (fx, fy, cx, cy) or S of them).
    Returns a list of `(new_pose 4x4, initial_error, final_error, outliers, n_outliers)`."""
    ctx = ctx or L.default_context()
    S = len(poses)
    px = [np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 2) for p in pixels]
    pt = [np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 3) for p in points]
    off = np.zeros(S + 1, dtype=np.int32); off[1:] = np.cumsum([len(p) for p in px])
    if [len(p) for p in px] != [len(p) for p in pt]:
        raise ValueError("pixels and points must have matching lengths")
    cams = np.asarray([c.intrinsics if hasattr(c, "intrinsics") else c for c in (cameras if np.ndim(cameras[0]) else [cameras] * S)], dtype=np.float64).reshape(S, 4)
    pin = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.float64).T for p in poses])) if S else np.zeros((0, 4, 4))   # column-major
    pxa = np.concatenate(px) if S else np.zeros((0, 2)); pta = np.concatenate(pt) if S else np.zeros((0, 3))
    out = np.zeros((S, 16)); e0 = np.zeros(S); e1 = np.zeros(S); no = np.zeros(S, dtype=np.int32)
    outl = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    ctx.check(ctx.lib.slam_pnp_ba_batch(ctx.h, S, L.ptr(off, L.i32p), L.ptr(cams), L.ptr(pin), L.ptr(pxa), L.ptr(pta), int(iters_fast), int(iterations),
                                        float(depth_eps), float(repr_eps), L.ptr(out), L.ptr(e0), L.ptr(e1), L.ptr(outl, L.u8p), L.ptr(no, L.i32p)))
    return [(out[z].reshape(4, 4).T.copy(), float(e0[z]), float(e1[z]), outl[off[z]:off[z + 1]].astype(bool), int(no[z])) for z in range(S)]
