"""Host mirror of `slam_frontend_*` (include/slamhip.h): one live stream, one C call per frame -- the per-frame work of the reference's
front-end task (src/front_end.jl:58-113, :454-470: preprocess! + klt_tracking!) and, at key-frames, extract_keypoints!
(src/map_manager.jl:98-113) plus the mapper's right pyramid, stereo matching and triangulate_stereo! (src/mapper.jl:51-66, :142-183).
The keypoint list lives in HBM (`FrontEnd.keypoints()` downloads it); `step` returns (frame index, list length)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .keypoint_set import KeypointSet, stream_params
from .triangulation import projection_matrices


class FrontEndConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("H", "W", "pyramid_levels", "pyramid_levels_3d", "window", "iterations", "max_points", "radius", "grid_rows",
                                          "grid_cols", "cell_size", "cap", "pyr_mode", "lookahead", "right_target_only", "reserved")] + \
               [(n, C.c_double) for n in ("eig_thr", "eps", "max_distance", "sigma_mask", "min_response", "epipolar_error", "max_error", "min_depth", "pyr_sigma")]


class FrontEnd:
    def __init__(self, shape, params, extractor, cap=None, fast=False, lookahead=False, right_target_only=True, device=0,
                 sigma_mask=3.0, min_response=1e-4, eig_thr=1e-4, eps=1e-2, max_error=3.0, min_depth=0.1, epipolar_error=2.0, pyramid_levels_3d=1):
        H, W = shape
        e = extractor
        cap = cap or (e.max_points + e.grid_resolution[0] * e.grid_resolution[1] + 64)
        self.cfg = FrontEndConfig(H=H, W=W, pyramid_levels=params.pyramid_levels, pyramid_levels_3d=pyramid_levels_3d, window=params.window_size,
                                  iterations=30, max_points=e.max_points, radius=e.radius, grid_rows=e.grid_resolution[0], grid_cols=e.grid_resolution[1],
                                  cell_size=e.cell_size, cap=cap, pyr_mode=3 if fast else 1, lookahead=1 if lookahead else 0,
                                  right_target_only=1 if right_target_only else 0, reserved=0, eig_thr=eig_thr, eps=eps, max_distance=params.max_ktl_distance,
                                  sigma_mask=sigma_mask, min_response=min_response,
                                  epipolar_error=epipolar_error,
                                  max_error=max_error, min_depth=min_depth, pyr_sigma=1.0)
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.slam_frontend_create(int(device), C.byref(self.cfg), C.byref(h))
        if rc:
            raise L.SlamHipError(f"slam_frontend_create: {rc}: {self.lib.slam_last_error(None).decode()}")
        self.h = h
        self.cap, self.shape = cap, (H, W)
        self._frame, self._count = C.c_int32(-1), C.c_int32(0)

    def close(self):
        if getattr(self, "h", None):
            self.lib.slam_frontend_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise L.SlamHipError(f"slam_frontend: {rc}: {self.lib.slam_frontend_last_error(self.h).decode()}")

    @staticmethod
    def tri_params(cam1, cam2, T21, Twc):
        """the 72 doubles of a key-frame's triangulation: P1, P2, T21 (column-major 4 x 4), cam1, cam2, Twc"""
        P1, P2 = projection_matrices(cam1, cam2, T21)
        cm = lambda M: np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(-1)
        return np.ascontiguousarray(np.concatenate([cm(P1), cm(P2), cm(T21), np.asarray(cam1, float), np.asarray(cam2, float), cm(Twc)]))

    def step(self, left_u8, right_u8=None, params=None, prior=2, stereo_params=None, stereo_prior=2, tri=None, cull_flags_dev=None):
        """feed one frame (column-major H x W bytes: `np.ascontiguousarray(img.T)` of a row-major image; right_u8 given = key-frame) and process the
        frame due (lookahead: the one before) -> (frame index or -1, list length)"""
        l = np.ascontiguousarray(left_u8, dtype=np.uint8); r = None if right_u8 is None else np.ascontiguousarray(right_u8, dtype=np.uint8)
        p = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
        sp = None if stereo_params is None else np.ascontiguousarray(stereo_params, dtype=np.float64)
        t = None if tri is None else np.ascontiguousarray(tri, dtype=np.float64)
        self._check(self.lib.slam_frontend_step(self.h, L.ptr(l, L.u8p), L.ptr(r, L.u8p) if r is not None else None, L.ptr(p) if p is not None else None, int(prior),
                                                L.ptr(sp) if sp is not None else None, int(stereo_prior), L.ptr(t) if t is not None else None,
                                                C.c_void_p(int(cull_flags_dev)) if cull_flags_dev else None, C.byref(self._frame), C.byref(self._count)))
        return self._frame.value, self._count.value

    def step_ptr(self, left_ptr, right_ptr, params_ptr, prior, stereo_params_ptr, stereo_prior, tri_ptr, cull_flags_dev=None):
        """the same with raw addresses (ints) prepared by the caller -- no array conversion per frame: what a C / Julia host passes"""
        self._check(self.lib.slam_frontend_step(self.h, C.cast(left_ptr, L.u8p), C.cast(right_ptr, L.u8p) if right_ptr else None,
                                                C.cast(params_ptr, L.f64p) if params_ptr else None, prior, C.cast(stereo_params_ptr, L.f64p) if stereo_params_ptr else None,
                                                stereo_prior, C.cast(tri_ptr, L.f64p) if tri_ptr else None, C.c_void_p(int(cull_flags_dev)) if cull_flags_dev else None,
                                                C.byref(self._frame), C.byref(self._count)))
        return self._frame.value, self._count.value

    def flush(self, params=None, prior=2, stereo_params=None, stereo_prior=2, tri=None, cull_flags_dev=None):
        p = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
        sp = None if stereo_params is None else np.ascontiguousarray(stereo_params, dtype=np.float64)
        t = None if tri is None else np.ascontiguousarray(tri, dtype=np.float64)
        self._check(self.lib.slam_frontend_flush(self.h, L.ptr(p) if p is not None else None, int(prior), L.ptr(sp) if sp is not None else None, int(stereo_prior),
                                                 L.ptr(t) if t is not None else None, C.c_void_p(int(cull_flags_dev)) if cull_flags_dev else None,
                                                 C.byref(self._frame), C.byref(self._count)))
        return self._frame.value, self._count.value

    def keypoints(self):
        """the stream's list: dict(yx, is_3d, xyz, ids, stereo_yx, has_stereo)"""
        ks = KeypointSet.__new__(KeypointSet)
        ks.S, ks.cap = 1, self.cap
        ks.h = C.c_void_p(self.lib.slam_frontend_keypoints(self.h))
        ks.ctx = _BorrowedCtx(self.lib, C.c_void_p(self.lib.slam_frontend_ctx(self.h)))
        try:
            return ks.download(0)
        finally:
            ks.h = None                                     # (borrowed: the front-end owns the set)


class _BorrowedCtx:
    """the front-end's own tracking context, for the calls that take one (not closed by the mirror)"""
    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def check(self, rc):
        if rc:
            raise L.SlamHipError(f"libslamhip error {rc}: {self.lib.slam_last_error(self.h).decode()}")
