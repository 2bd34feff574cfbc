"""Host mirror of the front-end's pose seam (reference: src/front_end.jl:132-219, compute_pose!).

`p3p_ransac(points, pixels, pdn_positions, K; threshold)` keeps the reference call's argument order and result
shape `(n_inliers, (KP, inliers, error))`; the hypotheses are generated and scored on the GPU (slam_p3p_ransac).
The sample triples are drawn on the host and handed over (the C ABI has no RNG): pass `samples` to reproduce a run."""
import ctypes as C

import numpy as np

from . import _lib as L


def draw_samples(n, iters, seed=0, k=3):
    """iters 0-based k-tuples of distinct indices out of n points (host-side stand-in for the RANSAC sampler)."""
    if n < k:
        return np.zeros((0, k), dtype=np.int32)
    if k != 3:
        rng = np.random.default_rng(seed)
        return np.argsort(rng.random((iters, n)), axis=1)[:, :k].astype(np.int32) if n <= 64 else _distinct(rng, n, iters, k)
    rng = np.random.default_rng(seed)
    s = rng.integers(0, n, size=(iters, 3), dtype=np.int64)
    s[:, 1] = (s[:, 0] + 1 + rng.integers(0, n - 1, iters)) % n                      # != s0
    third = rng.integers(0, n - 2, iters)
    lo, hi = np.minimum(s[:, 0], s[:, 1]), np.maximum(s[:, 0], s[:, 1])
    third = third + (third >= lo); third = third + (third >= hi)                    # skip both earlier picks
    s[:, 2] = third
    return s.astype(np.int32)


def _distinct(rng, n, iters, k):
    s = rng.integers(0, n, size=(iters, k))
    for _ in range(64):                                   # redraw the (rare) tuples with a repeated index
        srt = np.sort(s, axis=1)
        bad = (srt[:, 1:] == srt[:, :-1]).any(axis=1)
        if not bad.any():
            break
        s[bad] = rng.integers(0, n, size=(int(bad.sum()), k))
    return s.astype(np.int32)


def p3p_ransac(points, pixels_xy, pdn_positions, K, threshold=1.0, samples=None, iterations=256, seed=0, ctx=None,
               return_pose=False):
    """points (n, 3) map points, pixels_xy (n, 2) undistorted pixels in (x, y) order, pdn_positions (n, 3) bearing
    vectors, K 3x3.  Returns `(n_inliers, (KP, inliers, error))` like the reference, or None when no sample gave a
    pose (`res === nothing`, front_end.jl:168); with return_pose=True the model tuple also carries Rt = [R | t]."""
    ctx = ctx or L.default_context()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    px = np.ascontiguousarray(pixels_xy, dtype=np.float64).reshape(-1, 2)
    bd = np.ascontiguousarray(pdn_positions, dtype=np.float64).reshape(-1, 3)
    n = len(pts)
    if len(px) != n or len(bd) != n:
        raise ValueError("points, pixels and pdn_positions must have the same length")
    Kf = np.asfortranarray(K, dtype=np.float64)
    if Kf.shape != (3, 3):
        raise ValueError("K must be 3x3")
    sm = draw_samples(n, iterations, seed) if samples is None else np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 3)
    KP = np.zeros((3, 4), order="F"); Rt = np.zeros((3, 4), order="F")
    inl = np.zeros(n, dtype=np.uint8)
    cnt, bi, err = C.c_int(), C.c_int(), C.c_double()
    ctx.check(ctx.lib.slam_p3p_ransac(ctx.h, L.ptr(pts), L.ptr(px), L.ptr(bd), n, L.ptr(Kf), float(threshold),
                                      L.ptr(sm, L.i32p), len(sm), L.ptr(KP), L.ptr(Rt), L.ptr(inl, L.u8p),
                                      C.byref(cnt), C.cast(C.byref(err), L.f64p), C.byref(bi)))
    if cnt.value == 0:
        return None
    model = (np.array(KP), inl.view(np.bool_), err.value)
    if return_pose:
        model = model + (np.array(Rt), bi.value)
    return cnt.value, model


def five_point_ransac(previous_points_xy, current_points_xy, previous_pd, current_pd, K1, K2, max_repr_error=1.0,
                      samples=None, iterations=128, seed=0, ctx=None, return_extra=False):
    """five_point_ransac of compute_pose_5pt! (front_end.jl:305-308): pixels (n, 2) in (x, y) order, normalised
    coordinates (n, 2), K1 / K2 3x3.  Returns `(n_inliers, (E, P, inliers, error))` -- P = [R | t] 3x4, previous ->
    current, |t| = 1 -- with n_inliers = 0 and zero matrices when no sample gave a pose (the reference then fails its
    `n_inliers < 5` test, :309)."""
    ctx = ctx or L.default_context()
    a = np.ascontiguousarray(previous_points_xy, dtype=np.float64).reshape(-1, 2)
    b = np.ascontiguousarray(current_points_xy, dtype=np.float64).reshape(-1, 2)
    c = np.ascontiguousarray(previous_pd, dtype=np.float64).reshape(-1, 2)
    d = np.ascontiguousarray(current_pd, dtype=np.float64).reshape(-1, 2)
    n = len(a)
    if not (len(b) == len(c) == len(d) == n):
        raise ValueError("the four point lists must have the same length")
    k1 = np.asfortranarray(K1, dtype=np.float64); k2 = np.asfortranarray(K2, dtype=np.float64)
    if k1.shape != (3, 3) or k2.shape != (3, 3):
        raise ValueError("K1, K2 must be 3x3")
    sm = draw_samples(n, iterations, seed, k=5) if samples is None else np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 5)
    E = np.zeros((3, 3), order="F"); P = np.zeros((3, 4), order="F")
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    cnt, bi, err = C.c_int(), C.c_int(), C.c_double()
    ctx.check(ctx.lib.slam_five_point_ransac(ctx.h, L.ptr(a), L.ptr(b), L.ptr(c), L.ptr(d), n, L.ptr(k1), L.ptr(k2),
                                             float(max_repr_error), L.ptr(sm, L.i32p), len(sm), L.ptr(E), L.ptr(P),
                                             L.ptr(inl, L.u8p), C.byref(cnt), C.cast(C.byref(err), L.f64p), C.byref(bi)))
    model = (np.array(E), np.array(P), inl[:n].view(np.bool_), err.value)
    if return_extra:
        model = model + (bi.value,)
    return cnt.value, model


# ---- S lock-stepped streams: one set of launches for all of them -------------------------------------------------
def _concat(lists, width):
    arrs = [np.ascontiguousarray(a, dtype=np.float64).reshape(-1, width) for a in lists]
    off = np.zeros(len(arrs) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(a) for a in arrs])
    return (np.concatenate(arrs) if arrs else np.zeros((0, width))), off


def p3p_ransac_batch(points, pixels_xy, pdn_positions, K, threshold=1.0, samples=None, iterations=256, seed=0, ctx=None):
    """S problems at once (lists of per-stream arrays; K: one 3x3 or a list of S).  Returns a list of S results shaped
    like p3p_ransac(..., return_pose=True): `(n_inliers, (KP, inliers, error, Rt, best_iter))` or None."""
    ctx = ctx or L.default_context()
    S = len(points)
    pts, off = _concat(points, 3); px, off2 = _concat(pixels_xy, 2); bd, off3 = _concat(pdn_positions, 3)
    if not (np.array_equal(off, off2) and np.array_equal(off, off3)):
        raise ValueError("per-stream lists must have matching lengths")
    Ks = np.asarray(K, dtype=np.float64)
    Ks = np.broadcast_to(Ks, (S, 3, 3)) if Ks.ndim == 2 else Ks
    Kf = np.ascontiguousarray(np.transpose(Ks, (0, 2, 1)))                          # column-major per problem
    if samples is None:
        samples = [draw_samples(off[z + 1] - off[z], iterations, seed + z) for z in range(S)]
        samples = [s if len(s) else np.full((iterations, 3), -1, np.int32) for s in samples]
    sm = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.int32).reshape(-1, 3) for s in samples])) if S else np.zeros((0, 0, 3), np.int32)
    iters = sm.shape[1] if S else 0
    KP = np.zeros((S, 12)); Rt = np.zeros((S, 12)); inl = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    cnt = np.zeros(S, dtype=np.int32); bi = np.zeros(S, dtype=np.int32); err = np.zeros(S)
    ctx.check(ctx.lib.slam_p3p_ransac_batch(ctx.h, S, L.ptr(off, L.i32p), L.ptr(pts), L.ptr(px), L.ptr(bd), L.ptr(Kf), float(threshold),
                                            L.ptr(sm, L.i32p), iters, L.ptr(KP), L.ptr(Rt), L.ptr(inl, L.u8p), L.ptr(cnt, L.i32p),
                                            L.ptr(err), L.ptr(bi, L.i32p)))
    out = []
    for z in range(S):
        if cnt[z] == 0:
            out.append(None)
        else:
            out.append((int(cnt[z]), (KP[z].reshape(4, 3).T.copy(), inl[off[z]:off[z + 1]].view(np.bool_).copy(), float(err[z]),
                                      Rt[z].reshape(4, 3).T.copy(), int(bi[z]))))
    return out


def five_point_ransac_batch(previous_points_xy, current_points_xy, previous_pd, current_pd, K1, K2, max_repr_error=1.0,
                            samples=None, iterations=128, seed=0, ctx=None):
    """S problems at once; returns a list of `(n_inliers, (E, P, inliers, error, best_iter))`."""
    ctx = ctx or L.default_context()
    S = len(previous_points_xy)
    a, off = _concat(previous_points_xy, 2); b, o2 = _concat(current_points_xy, 2)
    c, o3 = _concat(previous_pd, 2); d, o4 = _concat(current_pd, 2)
    if not (np.array_equal(off, o2) and np.array_equal(off, o3) and np.array_equal(off, o4)):
        raise ValueError("per-stream lists must have matching lengths")
    def kk(K):
        Ks = np.asarray(K, dtype=np.float64)
        Ks = np.broadcast_to(Ks, (S, 3, 3)) if Ks.ndim == 2 else Ks
        return np.ascontiguousarray(np.transpose(Ks, (0, 2, 1)))
    k1, k2 = kk(K1), kk(K2)
    if samples is None:
        samples = [draw_samples(off[z + 1] - off[z], iterations, seed + z, k=5) for z in range(S)]
        samples = [s if len(s) else np.full((iterations, 5), -1, np.int32) for s in samples]
    sm = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.int32).reshape(-1, 5) for s in samples])) if S else np.zeros((0, 0, 5), np.int32)
    iters = sm.shape[1] if S else 0
    E = np.zeros((S, 9)); P = np.zeros((S, 12)); inl = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    cnt = np.zeros(S, dtype=np.int32); bi = np.zeros(S, dtype=np.int32); err = np.zeros(S)
    ctx.check(ctx.lib.slam_five_point_ransac_batch(ctx.h, S, L.ptr(off, L.i32p), L.ptr(a), L.ptr(b), L.ptr(c), L.ptr(d), L.ptr(k1), L.ptr(k2),
                                                   float(max_repr_error), L.ptr(sm, L.i32p), iters, L.ptr(E), L.ptr(P), L.ptr(inl, L.u8p),
                                                   L.ptr(cnt, L.i32p), L.ptr(err), L.ptr(bi, L.i32p)))
    return [(int(cnt[z]), (E[z].reshape(3, 3).T.copy(), P[z].reshape(4, 3).T.copy(), inl[off[z]:off[z + 1]].view(np.bool_).copy(),
                           float(err[z]), int(bi[z]))) for z in range(S)]
