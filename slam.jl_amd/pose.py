"""Host mirror of the front-end's pose seam (reference: src/front_end.jl:132-219, compute_pose!).

`p3p_ransac(points, pixels, pdn_positions, K; threshold)` keeps the reference call's argument order and result
shape `(n_inliers, (KP, inliers, error))`; the hypotheses are generated and scored on the GPU (slam_p3p_ransac).
The sample triples are drawn on the host and handed over (the C ABI has no RNG): pass `samples` to reproduce a run."""
import ctypes as C

import numpy as np

from . import _lib as L


def draw_samples(n, iters, seed=0):
    """iters distinct 0-based index triples out of n points (host-side stand-in for the RANSAC sampler)."""
    if n < 3:
        return np.zeros((0, 3), dtype=np.int32)
    rng = np.random.default_rng(seed)
    s = rng.integers(0, n, size=(iters, 3), dtype=np.int64)
    s[:, 1] = (s[:, 0] + 1 + rng.integers(0, n - 1, iters)) % n                      # != s0
    third = rng.integers(0, n - 2, iters)
    lo, hi = np.minimum(s[:, 0], s[:, 1]), np.maximum(s[:, 0], s[:, 1])
    third = third + (third >= lo); third = third + (third >= hi)                    # skip both earlier picks
    s[:, 2] = third
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
