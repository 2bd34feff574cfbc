"""Extractor mirror (reference: src/extractor.jl)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


@dataclass
class Extractor:
    """extractor.jl:7-22; built in SlamManager (SLAM.jl:149-160)."""
    max_points: int
    radius: int
    grid_resolution: tuple
    cell_size: int

    @classmethod
    def from_params(cls, params, camera):
        radius = max(5, params.max_distance // 2)                       # SLAM.jl:149
        grid = (-(-camera.height // params.max_distance), -(-camera.width // params.max_distance))  # SLAM.jl:150-151
        return cls(params.max_nb_keypoints, radius, grid, params.max_distance)


def _cap(e, n_cur):
    n_cells = e.grid_resolution[0] * e.grid_resolution[1]
    k = -(-max(e.max_points - n_cur, 0) // n_cells)
    return n_cells * max(k, 1)


def detect(e, image, current_points, sigma_mask=3.0, min_response=1e-4, ctx=None):
    """detect(e::Extractor, image, current_points; σ_mask) -> (n, 2) int64 (row, col), 1-based.

    `image` is an H x W float64 array (any order; passed in Julia's column-major
    layout) or an LKPyramid, in which case its device-resident base layer is used."""
    from .optical_flow import LKPyramid
    ctx = ctx or L.default_context()
    cur = np.ascontiguousarray(current_points, dtype=np.float64).reshape(-1, 2)
    cap = _cap(e, len(cur))
    out = np.zeros((cap, 2), dtype=np.int64)
    n = C.c_int(0)
    if isinstance(image, LKPyramid):
        rc = ctx.lib.slam_detect_pyr(ctx.h, image.h, L.ptr(cur), len(cur), e.max_points, e.radius,
                                     e.grid_resolution[0], e.grid_resolution[1], e.cell_size,
                                     float(sigma_mask), float(min_response), L.ptr(out, L.i64p), cap, C.byref(n))
    else:
        img = np.asfortranarray(image, dtype=np.float64)
        H, W = img.shape
        rc = ctx.lib.slam_detect(ctx.h, L.ptr(img), H, W, L.ptr(cur), len(cur), e.max_points, e.radius,
                                 e.grid_resolution[0], e.grid_resolution[1], e.cell_size,
                                 float(sigma_mask), float(min_response), L.ptr(out, L.i64p), cap, C.byref(n))
    ctx.check(rc)
    return out[:n.value].copy()


def detect_batch(e, batch, current_points, stream_index, sigma_mask=3.0, min_response=1e-4, ctx=None):
    """detect() for every stream of a PyramidBatch in one launch (slam_detect_batch).

    current_points (n, 2) with stream_index (n,) sorted ascending (points of stream s contiguous).
    Returns (keypoints (m, 2) int64, stream_index (m,) int32), grouped by stream."""
    ctx = ctx or batch.ctx
    S = batch.S
    cur = np.ascontiguousarray(current_points, dtype=np.float64).reshape(-1, 2)
    sid = np.asarray(stream_index, dtype=np.int64)
    if len(sid) > 1 and np.any(np.diff(sid) < 0):
        raise ValueError("detect_batch: current_points must be grouped by ascending stream_index")
    cur_off = np.zeros(S + 1, dtype=np.int32)
    cur_off[1:] = np.cumsum(np.bincount(sid, minlength=S)[:S])
    cap = S * _cap(e, 0)
    out = np.empty((cap, 2), dtype=np.int64)
    out_off = np.zeros(S + 1, dtype=np.int32)
    ctx.check(ctx.lib.slam_detect_batch(ctx.h, batch.pyramids[0].h, S, L.ptr(cur), L.ptr(cur_off, L.i32p), e.max_points, e.radius,
                                        e.grid_resolution[0], e.grid_resolution[1], e.cell_size, float(sigma_mask),
                                        float(min_response), L.ptr(out, L.i64p), cap, L.ptr(out_off, L.i32p)))
    m = int(out_off[S])
    return out[:m].copy(), np.repeat(np.arange(S, dtype=np.int32), np.diff(out_off))


def brief_pattern(size=256, window=9, seed=123):
    """A BRIEF sampling table (size x 4: dy1, dx1, dy2, dx2), Gaussian sampling
    N(0, window^2/25) clipped to the window like ImageFeatures' gaussian
    sampler.  ImageFeatures draws it from Julia's RNG (seed 123), which cannot
    be reproduced outside Julia: the Julia shim passes ITS table through the same
    argument, this one only serves callers without Julia."""
    rng = np.random.default_rng(seed)
    lim = window // 2
    s = np.clip(np.rint(rng.normal(0.0, window / 5.0, (size, 4))), -lim, lim)
    return s.astype(np.int32)


def describe(e, image, keypoints, pattern=None, sigma=np.sqrt(2.0), window=9, ctx=None):
    """describe(e, image, keypoints) -> (descriptors (n', 4) uint64, keypoints (n', 2) int64)."""
    ctx = ctx or L.default_context()
    img = np.asfortranarray(image, dtype=np.float64)
    H, W = img.shape
    rc_in = np.ascontiguousarray(keypoints, dtype=np.int64).reshape(-1, 2)
    pat = np.ascontiguousarray(brief_pattern() if pattern is None else pattern, dtype=np.int32).reshape(-1, 4)
    nb = len(pat)
    bits = np.zeros((len(rc_in), nb // 64), dtype=np.uint64)
    out_rc = np.zeros((len(rc_in), 2), dtype=np.int64)
    n = C.c_int(0)
    ctx.check(ctx.lib.slam_describe(ctx.h, L.ptr(img), H, W, L.ptr(rc_in, L.i64p), len(rc_in), L.ptr(pat, L.i32p), nb,
                                    float(sigma), int(window), L.ptr(bits, L.u64p), L.ptr(out_rc, L.i64p), C.byref(n)))
    return bits[:n.value].copy(), out_rc[:n.value].copy()
