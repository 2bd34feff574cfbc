"""Host mirror of the mapper's triangulation seam (reference: src/mapper.jl:142-262).

`triangulate(...)` is the array-level body of triangulate_stereo! / triangulate_temporal!: the per-keypoint DLT
triangulation and the depth / reprojection gates run on the GPU (slam_triangulate); the map surgery that follows
(update_mappoint!, remove_stereo_keypoint!, remove_mappoint_obs!) is the caller's, driven by the returned status."""
import numpy as np

from . import _lib as L


def projection_matrices(cam1, cam2, T21):
    """P1 = to_4x4(K1) * I, P2 = to_4x4(K2) * T21 (mapper.jl:151-152, 194-197, 233)."""
    def k4(c):
        fx, fy, cx, cy = c
        return np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    return k4(cam1), k4(cam2) @ np.asarray(T21, dtype=np.float64)


def triangulate(cam1, cam2, T21, px1_yx, px2_yx, max_error, min_depth=0.1, parallax=None, min_parallax=20.0, ctx=None):
    """cam = (fx, fy, cx, cy); T21: 4x4 camera-1 -> camera-2 transform; pixels (n, 2) (y, x), undistorted.
    parallax=None: stereo semantics.  Returns (xyz (n, 3) in camera-1 coordinates, status (n,) bool)."""
    ctx = ctx or L.default_context()
    P1, P2 = projection_matrices(cam1, cam2, T21)
    P1 = np.asfortranarray(P1); P2 = np.asfortranarray(P2); T = np.asfortranarray(T21, dtype=np.float64)
    c1 = np.ascontiguousarray(cam1, dtype=np.float64); c2 = np.ascontiguousarray(cam2, dtype=np.float64)
    a = np.ascontiguousarray(px1_yx, dtype=np.float64).reshape(-1, 2); b = np.ascontiguousarray(px2_yx, dtype=np.float64).reshape(-1, 2)
    n = len(a)
    out = np.zeros((n, 3)); st = np.zeros(n, dtype=np.uint8)
    par = None if parallax is None else np.ascontiguousarray(parallax, dtype=np.float64)
    ctx.check(ctx.lib.slam_triangulate(ctx.h, L.ptr(P1), L.ptr(P2), L.ptr(T), L.ptr(c1), L.ptr(c2), L.ptr(a), L.ptr(b), n,
                                       float(max_error), float(min_depth), L.ptr(par) if par is not None else None, float(min_parallax),
                                       L.ptr(out), L.ptr(st, L.u8p)))
    return out, st.view(np.bool_)
