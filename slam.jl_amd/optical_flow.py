"""LKPyramid / LucasKanade / fb_tracking! mirrors (reference:
src/optical_flow/pyramid.jl, src/optical_flow/lucas_kanade.jl, src/tracker.jl)
and the array-level protocol of optical_flow_matching! (src/map_manager.jl:451-564)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L

PLANES = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")


@dataclass
class LucasKanade:
    """lucas_kanade.jl:1-7"""
    iterations: int = 30
    window_size: int = 9
    pyramid_levels: int = 3
    eigenvalue_threshold: float = 1e-4
    eps: float = 1e-2


class LKPyramid:
    """Opaque device-resident pyramid (pyramid.jl:16-24): nobody outside
    optical_flow/ reads the planes in the reference, so a handle is enough.

    LKPyramid(image, levels; σ=1.0) builds with CONSTRUCTOR semantics
    (pyramid.jl:40-79); update_(lk, img) rebuilds with update! semantics."""

    def __init__(self, image=None, levels=3, sigma=1.0, shape=None, ctx=None, _handle=None):
        self.ctx = ctx or L.default_context()
        if _handle is not None:
            self.h = _handle
        else:
            if image is not None:
                image = np.asfortranarray(image, dtype=np.float64)
                shape = image.shape
            h = C.c_void_p()
            self.ctx.check(self.ctx.lib.slam_pyr_create(self.ctx.h, shape[0], shape[1], levels, C.byref(h)))
            self.h = h
            if image is not None:
                self.ctx.check(self.ctx.lib.slam_pyr_update(self.ctx.h, self.h, L.ptr(image), 0, float(sigma)))
        self.levels = self.ctx.lib.slam_pyr_levels(self.h)

    def level_shape(self, level):
        H, W = C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.slam_pyr_shape(self.h, level, C.byref(H), C.byref(W)))
        return H.value, W.value

    def plane(self, name, level, ctx=None):
        """Download one plane (for parity tests); level 0-based; H x W Fortran array.  `ctx`: the context whose stream does the copy."""
        H, W = self.level_shape(level)
        out = np.empty((H, W), order="F")
        c = ctx or self.ctx
        c.check(c.lib.slam_pyr_download(c.h, self.h, PLANES.index(name), level, L.ptr(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.slam_pyr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def has_gradients(lk):
    return True


def update_(lk, img, sigma=1.0, device_ptr=None, sync=True, ctx=None, fast=False, target_only=False, chain=False):
    """update!(lk, img; σ) pyramid.jl:81-96.  `device_ptr`: image already in HBM;
    `ctx`: enqueue on another context's stream (default: the pyramid's own);
    `fast`: segmented recurrences (mode 3) -- planes agree with the sequential,
    bit-exact mode to ~1e-13 relative instead of bit for bit."""
    mode = (3 if fast else 1) | (16 if target_only else 0) | (32 if chain else 0)      # SLAM_PYR_TARGET_ONLY: only ever matched INTO (the mapper's right pyramid); SLAM_PYR_CHAIN: unforked replay
    if device_ptr is not None:
        c = ctx or lk.ctx
        c.check(c.lib.slam_pyr_update_dev(c.h, lk.h, C.c_void_p(device_ptr), mode, float(sigma), 1 if sync else 0))
    elif np.asarray(img).dtype == np.uint8:
        img = np.asfortranarray(img)                     # 8-bit frame: converted to Gray{Float64} (raw/255) on the device
        assert img.shape == lk.level_shape(0)
        lk.ctx.check(lk.ctx.lib.slam_pyr_update_u8(lk.ctx.h, lk.h, L.ptr(img, L.u8p), mode, float(sigma)))
    else:
        img = np.asfortranarray(img, dtype=np.float64)
        assert img.shape == lk.level_shape(0)
        lk.ctx.check(lk.ctx.lib.slam_pyr_update(lk.ctx.h, lk.h, L.ptr(img), mode, float(sigma)))
    return lk


def copy_(dst, src):
    """copy!(dst, src) pyramid.jl:28-38"""
    dst.ctx.check(dst.ctx.lib.slam_pyr_copy(dst.ctx.h, dst.h, src.h))
    return dst


def deepcopy(lk):
    """deepcopy(current_pyramid), SLAM.jl:218"""
    h = C.c_void_p()
    lk.ctx.check(lk.ctx.lib.slam_pyr_clone(lk.ctx.h, lk.h, C.byref(h)))
    return LKPyramid(ctx=lk.ctx, _handle=h)


def fb_tracking_(previous_pyramid, current_pyramid, keypoints, displacement=None,
                 iterations=30, window_size=11, pyramid_levels=3, max_distance=0.5,
                 eigenvalue_threshold=1e-4, eps=1e-2, ctx=None):
    """fb_tracking!(prev, cur, keypoints; displacement, iterations, window_size,
    pyramid_levels, max_distance) -> (new_keypoints (n,2), status (n,) bool) -- tracker.jl:70-82.
    Returns None for an empty keypoint list (tracker.jl:24)."""
    ctx = ctx or previous_pyramid.ctx
    pts = np.ascontiguousarray(keypoints, dtype=np.float64).reshape(-1, 2)
    n = len(pts)
    if n == 0:
        return None
    d0 = None if displacement is None else np.ascontiguousarray(displacement, dtype=np.float64).reshape(-1, 2)
    out = np.empty((n, 2))
    status = np.zeros(n, dtype=np.uint8)
    rc = ctx.lib.slam_fb_track(ctx.h, previous_pyramid.h, current_pyramid.h, L.ptr(pts), L.ptr(d0), n,
                               pyramid_levels, window_size, iterations, float(eigenvalue_threshold), float(eps),
                               float(max_distance), L.ptr(out), L.ptr(status, L.u8p))
    if rc == -3:
        raise RuntimeError("Not enough layers in pyramids.")      # lucas_kanade.jl:15
    ctx.check(rc)
    return out, status.astype(bool)


def optical_flow_matching(from_pyramid, to_pyramid, pixels, is_3d, projections, params, pyramid_levels_3d=1,
                          iterations=30, fused=True, ctx=None):
    """Array-level protocol of optical_flow_matching! (map_manager.jl:451-564):
    3-D keypoints are tracked first with the projected prior on
    `pyramid_levels_3d` levels; the ones that fail join the 2-D keypoints and are
    tracked without prior on params.pyramid_levels levels.
    Returns (new_pixels (n,2), status (n,) bool).

    fused=True issues ONE slam_flow_match launch; fused=False issues the
    reference's two fb_tracking! calls (kept for the parity test: identical results)."""
    pixels = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    n = len(pixels)
    is_3d = np.ascontiguousarray(is_3d, dtype=np.uint8)
    proj = np.ascontiguousarray(projections, dtype=np.float64).reshape(-1, 2)
    if fused:
        ctx = ctx or from_pyramid.ctx
        if n == 0:
            return pixels.copy(), np.zeros(0, dtype=bool)
        out = np.empty((n, 2)); status = np.zeros(n, dtype=np.uint8)
        rc = ctx.lib.slam_flow_match(ctx.h, from_pyramid.h, to_pyramid.h, L.ptr(pixels), L.ptr(is_3d, L.u8p), L.ptr(proj), n,
                                     params.pyramid_levels, pyramid_levels_3d, params.window_size, iterations, 1e-4, 1e-2,
                                     float(params.max_ktl_distance), L.ptr(out), L.ptr(status, L.u8p))
        if rc == -3:
            raise RuntimeError("Not enough layers in pyramids.")
        ctx.check(rc)
        st = status.view(np.bool_)
        return np.where(st[:, None], out, pixels), st
    is3 = is_3d.astype(bool)
    new = pixels.copy()
    status = np.zeros(n, dtype=bool)
    ids3 = np.where(is3)[0]
    ids2 = list(np.where(~is3)[0])
    scale = 1.0 / 2.0 ** pyramid_levels_3d
    if len(ids3):
        disp = scale * (proj[ids3] - pixels[ids3])                                                       # map_manager.jl:494,504
        nk, st = fb_tracking_(from_pyramid, to_pyramid, pixels[ids3], displacement=disp, iterations=iterations,
                              pyramid_levels=pyramid_levels_3d, window_size=params.window_size,
                              max_distance=params.max_ktl_distance)
        ok = ids3[st]
        new[ok] = nk[st]; status[ok] = True
        ids2 += list(ids3[~st])                                                                        # map_manager.jl:533-538
    if len(ids2):
        ids2 = np.asarray(ids2)
        nk, st = fb_tracking_(from_pyramid, to_pyramid, pixels[ids2], iterations=iterations, pyramid_levels=params.pyramid_levels,
                              window_size=params.window_size, max_distance=params.max_ktl_distance)
        ok = ids2[st]
        new[ok] = nk[st]; status[ok] = True
    return new, status


def undistort_point(cam, dist, p_yx):
    """undistort_point (camera.jl:98-125) for an (n, 2) array of (y, x) pixels; cam = (fx, fy, cx, cy), dist = (k1, k2, p1, p2)."""
    fx, fy, cx, cy = cam
    k1, k2, p1, p2 = dist
    p = np.asarray(p_yx, dtype=np.float64).reshape(-1, 2)
    ny = (p[:, 0] - cy) / fy; nx = (p[:, 1] - cx) / fx
    s0 = ny * ny; s1 = nx * nx
    r2 = s0 + s1
    rd = 1.0 + k1 * r2 + k2 * r2 ** 2
    pr = ny * nx
    dtx = 2 * p1 * pr + p2 * (r2 + 2 * s0)
    dty = p1 * (r2 + 2 * s1) + 2 * p2 * pr
    return np.stack([(rd * ny + dty) * fy + cy, (rd * nx + dtx) * fx + cx], axis=1)


def optical_flow_matching_frame(from_pyramid, to_pyramid, pixels, is_3d, projections, params, image_size, stereo=False,
                                undistorted_left=None, right_cam=None, right_dist=(0.0, 0.0, 0.0, 0.0),
                                epipolar_error=2.0, ctx=None):
    """optical_flow_matching!(map_manager, frame, from, to, stereo) on the frame's keypoint arrays, with the gates the
    reference applies around the two fb_tracking! calls (map_manager.jl:451-564): 3-D keypoints whose projection is
    outside the target image are skipped (temporal, :501-506) or removed (stereo, :491-498); a stereo match must pass
    maybe_stereo_update! (:579-590: row difference of the undistorted pixels <= epipolar_error; the row of the left
    keypoint is kept).  The tracking itself is ONE slam_flow_match launch.
    Returns dict(new_pixels, updated, removed) over the input keypoints."""
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    n = len(px)
    is3 = np.asarray(is_3d).astype(bool)
    proj = np.ascontiguousarray(projections, dtype=np.float64).reshape(-1, 2)
    Himg, Wimg = image_size
    inside = (proj[:, 0] >= 1) & (proj[:, 0] <= Himg) & (proj[:, 1] >= 1) & (proj[:, 1] <= Wimg)      # camera.jl:91
    skipped = is3 & ~inside
    sel = np.where(~skipped)[0]
    new = px.copy(); updated = np.zeros(n, bool); removed = np.zeros(n, bool)
    if stereo:
        removed[skipped] = True
    if len(sel):
        out, st = optical_flow_matching(from_pyramid, to_pyramid, px[sel], is3[sel], proj[sel], params, ctx=ctx)
        if stereo:
            right_pixel = undistort_point(right_cam, right_dist, out)
            ok = st & ~(np.abs(np.asarray(undistorted_left)[sel, 0] - right_pixel[:, 0]) > epipolar_error)
            good = sel[ok]
            new[good, 1] = out[ok, 1]                                                                 # :587: row of the left keypoint kept
            updated[good] = True
        else:
            new[sel[st]] = out[st]; updated[sel[st]] = True
            removed[sel[~st]] = True                                                                  # :559
    return dict(new_pixels=new, updated=updated, removed=removed)


class PyramidBatch:
    """S pyramids backed by one allocation (slam_pyr_create_batch): every per-image kernel of the build takes the
    image index from grid.z, so S independent images cost one launch set.  `self.pyramids[s]` are ordinary LKPyramid
    handles."""

    def __init__(self, shape, levels=3, S=2, ctx=None):
        self.ctx = ctx or L.default_context()
        self.S = S
        hs = (C.c_void_p * S)()
        self.ctx.check(self.ctx.lib.slam_pyr_create_batch(self.ctx.h, shape[0], shape[1], levels, S, hs))
        self.pyramids = [LKPyramid(ctx=self.ctx, _handle=C.c_void_p(hs[s])) for s in range(S)]
        self._handles = (C.c_void_p * S)(*[p.h for p in self.pyramids])

    def update_(self, device_ptrs, sigma=1.0, sync=True, fast=False, ctx=None, u8=False, target_only=False):
        """update!() of all S pyramids from S device-resident images (list of device pointers to column-major
        H x W Float64 images, or to 8-bit frames when u8=True: converted raw / 255 on the device)."""
        c = ctx or self.ctx
        imgs = (C.c_void_p * self.S)(*[C.c_void_p(p) for p in device_ptrs])
        fn = c.lib.slam_pyr_update_batch_u8_dev if u8 else c.lib.slam_pyr_update_batch_dev
        # target_only (SLAM_PYR_TARGET_ONLY): the batch will only be matched INTO (the right frames of a stereo match): the coarser
        # levels get their layers only
        c.check(fn(c.h, self._handles, imgs, self.S, (3 if fast else 1) | (16 if target_only else 0), float(sigma), 1 if sync else 0))
        return self


def optical_flow_matching_batch(from_batch, to_batch, stream_index, pixels, is_3d, projections, params,
                                pyramid_levels_3d=1, iterations=30, ctx=None, status_only=False):
    """optical_flow_matching! for S lock-stepped streams in one launch (slam_flow_match_batch): point i belongs to
    stream stream_index[i]; pyramids from_batch.pyramids[s] -> to_batch.pyramids[s].  Returns (new_pixels, status)."""
    ctx = ctx or from_batch.ctx
    pixels = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    n = len(pixels)
    if n == 0:
        return pixels.copy(), np.zeros(0, dtype=bool)
    idx = np.ascontiguousarray(stream_index, dtype=np.int32)
    is3 = np.ascontiguousarray(is_3d, dtype=np.uint8)
    proj = np.ascontiguousarray(projections, dtype=np.float64).reshape(-1, 2)
    out = np.empty((n, 2)); status = np.zeros(n, dtype=np.uint8)
    rc = ctx.lib.slam_flow_match_batch(ctx.h, from_batch.pyramids[0].h, to_batch.pyramids[0].h, from_batch.S, L.ptr(idx, L.i32p),
                                       L.ptr(pixels), L.ptr(is3, L.u8p), L.ptr(proj), n, params.pyramid_levels, pyramid_levels_3d,
                                       params.window_size, iterations, 1e-4, 1e-2, float(params.max_ktl_distance),
                                       L.ptr(out), L.ptr(status, L.u8p))
    if rc == -3:
        raise RuntimeError("Not enough layers in pyramids.")
    ctx.check(rc)
    st = status.view(np.bool_)
    if status_only:                                   # e.g. stereo matching: only the flags are used (mapper.jl:58-66)
        return None, st
    return np.where(st[:, None], out, pixels), st


def optical_flow_matching_batch_kept(from_batch, to_batch, stream_index, pixels, is_3d, projections, params,
                                     pyramid_levels_3d=1, iterations=30, ctx=None):
    """optical_flow_matching_batch + removal of the keypoints whose tracking failed (slam_flow_match_batch_kept): returns
    (new_pixels (k, 2), is_3d (k,) bool, stream_index (k,) int32, source_index (k,) int32) of the survivors, in input order."""
    ctx = ctx or from_batch.ctx
    pixels = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    n = len(pixels)
    if n == 0:
        return pixels.copy(), np.zeros(0, dtype=bool), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)
    idx = np.ascontiguousarray(stream_index, dtype=np.int32)
    is3 = np.ascontiguousarray(is_3d, dtype=np.uint8) if np.asarray(is_3d).dtype != np.bool_ else np.ascontiguousarray(is_3d).view(np.uint8)
    proj = np.ascontiguousarray(projections, dtype=np.float64).reshape(-1, 2)
    out = np.empty((n, 2)); k3 = np.empty(n, dtype=np.uint8); kimg = np.empty(n, dtype=np.int32); ksrc = np.empty(n, dtype=np.int32)
    nk = C.c_int(0)
    rc = ctx.lib.slam_flow_match_batch_kept(ctx.h, from_batch.pyramids[0].h, to_batch.pyramids[0].h, from_batch.S, L.ptr(idx, L.i32p),
                                            L.ptr(pixels), L.ptr(is3, L.u8p), L.ptr(proj), n, params.pyramid_levels, pyramid_levels_3d,
                                            params.window_size, iterations, 1e-4, 1e-2, float(params.max_ktl_distance),
                                            L.ptr(out), L.ptr(k3, L.u8p), L.ptr(kimg, L.i32p), L.ptr(ksrc, L.i32p), C.byref(nk), None)
    if rc == -3:
        raise RuntimeError("Not enough layers in pyramids.")
    ctx.check(rc)
    k = nk.value
    return out[:k], k3[:k].view(np.bool_), kimg[:k], ksrc[:k]
