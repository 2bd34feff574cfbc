"""Device-resident keypoint lists of S lock-stepped streams (slam_kpset, include/slamhip.h): the arrays behind
Frame.keypoints (src/frame.jl) for the calls of the front-end / mapper hot path -- optical_flow_matching!
(src/map_manager.jl:451-564), extract_keypoints! (:98-113), triangulate_stereo! (src/mapper.jl:142-183) -- kept in HBM
between calls.  Only `counts()`, `upload()` and `download()` touch the host."""
import ctypes as C

import numpy as np

from . import _lib as L


def stream_params(S, Tcw=None, cam=None, dist=None, shift_yx=None):
    """S x 32 per-stream call parameters: [0..15] Tcw (column-major), [16..19] fx fy cx cy, [20..23] k1 k2 p1 p2, [24..25] shift."""
    p = np.zeros((S, 32))
    p[:, 16:18] = 1.0
    if Tcw is not None:
        T = np.asarray(Tcw, dtype=np.float64).reshape(-1, 4, 4)
        p[:, :16] = np.broadcast_to(T, (S, 4, 4)).transpose(0, 2, 1).reshape(S, 16)       # column-major
    if cam is not None:
        p[:, 16:20] = np.asarray(cam, dtype=np.float64)
    if dist is not None:
        p[:, 20:24] = np.asarray(dist, dtype=np.float64)
    if shift_yx is not None:
        p[:, 24:26] = np.asarray(shift_yx, dtype=np.float64).reshape(-1, 2)
    return np.ascontiguousarray(p)


class KeypointSet:
    def __init__(self, S, cap, ctx=None):
        self.ctx = ctx or L.default_context()
        self.S, self.cap = S, cap
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.slam_kpset_create(self.ctx.h, S, cap, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.slam_kpset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host <-> device (initialisation, tests, host consumers) ----
    def upload(self, s, yx, is_3d, xyz=None, ids=None, ctx=None):
        c = ctx or self.ctx
        yx = np.ascontiguousarray(yx, dtype=np.float64).reshape(-1, 2)
        f = np.ascontiguousarray(np.asarray(is_3d).astype(np.uint8))
        x = None if xyz is None else np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        i = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        c.check(c.lib.slam_kpset_upload(c.h, self.h, s, L.ptr(yx), L.ptr(f, L.u8p), L.ptr(x) if x is not None else None,
                                        L.ptr(i, L.i64p) if i is not None else None, len(yx)))

    def download(self, s, ctx=None):
        """dict(yx, is_3d, xyz, ids, stereo_yx, has_stereo) of stream s"""
        c = ctx or self.ctx
        cap = self.cap
        yx = np.empty((cap, 2)); f = np.empty(cap, np.uint8); xyz = np.empty((cap, 3)); ids = np.empty(cap, np.int64)
        syx = np.empty((cap, 2)); hs = np.empty(cap, np.uint8); n = C.c_int(0)
        c.check(c.lib.slam_kpset_download(c.h, self.h, s, L.ptr(yx), L.ptr(f, L.u8p), L.ptr(xyz), L.ptr(ids, L.i64p), L.ptr(syx),
                                          L.ptr(hs, L.u8p), cap, C.byref(n)))
        k = n.value
        return dict(yx=yx[:k].copy(), is_3d=f[:k].astype(bool), xyz=xyz[:k].copy(), ids=ids[:k].copy(), stereo_yx=syx[:k].copy(),
                    has_stereo=hs[:k].astype(bool))

    def counts(self, ctx=None):
        """the S list lengths: the one small device -> host copy of a step (synchronises the context's stream)"""
        c = ctx or self.ctx
        out = np.zeros(self.S, dtype=np.int32)
        c.check(c.lib.slam_kpset_counts(c.h, self.h, L.ptr(out, L.i32p)))
        return out

    # ---- enqueue-only calls ----
    def flow_match(self, from_batch, to_batch, params, stream_params_=None, prior=0, pyramid_levels_3d=1, iterations=30, n_bound=0, ctx=None):
        c = ctx or self.ctx
        sp = None if stream_params_ is None else np.ascontiguousarray(stream_params_, dtype=np.float64)
        rc = c.lib.slam_kpset_flow_match(c.h, self.h, from_batch.pyramids[0].h, to_batch.pyramids[0].h, L.ptr(sp) if sp is not None else None,
                                         prior, params.pyramid_levels, pyramid_levels_3d, params.window_size, iterations, 1e-4, 1e-2,
                                         float(params.max_ktl_distance), int(n_bound))
        if rc == -3:
            raise RuntimeError("Not enough layers in pyramids.")
        c.check(rc)

    def stereo_match(self, left_batch, right_batch, params, stream_params_=None, prior=0, pyramid_levels_3d=1, iterations=30,
                     epipolar_error=2.0, n_bound=0, ctx=None):
        c = ctx or self.ctx
        sp = None if stream_params_ is None else np.ascontiguousarray(stream_params_, dtype=np.float64)
        rc = c.lib.slam_kpset_stereo_match(c.h, self.h, left_batch.pyramids[0].h, right_batch.pyramids[0].h, L.ptr(sp) if sp is not None else None,
                                           prior, params.pyramid_levels, pyramid_levels_3d, params.window_size, iterations, 1e-4, 1e-2,
                                           float(params.max_ktl_distance), float(epipolar_error), int(n_bound))
        if rc == -3:
            raise RuntimeError("Not enough layers in pyramids.")
        c.check(rc)

    def remove(self, flags_dev_ptr, ctx=None):
        """flags_dev_ptr: device pointer to S x cap bytes (1 = remove), e.g. torch_tensor.data_ptr()"""
        c = ctx or self.ctx
        c.check(c.lib.slam_kpset_remove(c.h, self.h, C.c_void_p(flags_dev_ptr)))

    def detect(self, e, batch, sigma_mask=3.0, min_response=1e-4, ctx=None):
        c = ctx or self.ctx
        c.check(c.lib.slam_kpset_detect(c.h, self.h, batch.pyramids[0].h, e.max_points, e.radius, e.grid_resolution[0], e.grid_resolution[1],
                                        e.cell_size, float(sigma_mask), float(min_response)))

    def triangulate(self, cam1, cam2, T21, Twc, max_error, min_depth=0.1, n_bound=0, ctx=None):
        from .triangulation import projection_matrices
        c = ctx or self.ctx
        P1, P2 = projection_matrices(cam1, cam2, T21)
        P1 = np.asfortranarray(P1); P2 = np.asfortranarray(P2); T = np.asfortranarray(T21, dtype=np.float64)
        c1 = np.ascontiguousarray(cam1, dtype=np.float64); c2 = np.ascontiguousarray(cam2, dtype=np.float64)
        W = np.asarray(Twc, dtype=np.float64).reshape(-1, 4, 4)
        W = np.ascontiguousarray(np.broadcast_to(W, (self.S, 4, 4)).transpose(0, 2, 1).reshape(self.S, 16))
        c.check(c.lib.slam_kpset_triangulate(c.h, self.h, L.ptr(P1), L.ptr(P2), L.ptr(T), L.ptr(c1), L.ptr(c2), L.ptr(W),
                                             float(max_error), float(min_depth), int(n_bound)))

    def keyframe(self, ctx=None):
        """create_keyframe! for the lists: the current positions become the previous key-frame's observations (slam_kpset_keyframe)"""
        c = ctx or self.ctx
        c.check(c.lib.slam_kpset_keyframe(c.h, self.h))

    def upload_keyframe(self, s, kyx, has_kf, ctx=None):
        c = ctx or self.ctx
        k = np.ascontiguousarray(kyx, dtype=np.float64).reshape(-1, 2)
        f = np.ascontiguousarray(np.asarray(has_kf).astype(np.uint8))
        c.check(c.lib.slam_kpset_upload_keyframe(c.h, self.h, s, L.ptr(k), L.ptr(f, L.u8p), len(k)))

    def download_keyframe(self, s, ctx=None):
        c = ctx or self.ctx
        k = np.zeros((self.cap, 2)); f = np.zeros(self.cap, dtype=np.uint8); n = C.c_int(0)
        c.check(c.lib.slam_kpset_download_keyframe(c.h, self.h, s, L.ptr(k), L.ptr(f, L.u8p), self.cap, C.byref(n)))
        return k[:n.value].copy(), f[:n.value].astype(bool)

    def upload_first(self, s, first_yx, first_kf, kf_count, ctx=None):
        c = ctx or self.ctx
        f = np.ascontiguousarray(first_yx, dtype=np.float64).reshape(-1, 2)
        k = np.ascontiguousarray(first_kf, dtype=np.int32)
        c.check(c.lib.slam_kpset_upload_first(c.h, self.h, s, L.ptr(f), L.ptr(k, L.i32p), len(f), int(kf_count)))

    def download_first(self, s, ctx=None):
        """(first_yx, first_kf, key-frame counter) of stream s: where and by which key-frame each keypoint was first observed"""
        c = ctx or self.ctx
        f = np.zeros((self.cap, 2)); k = np.zeros(self.cap, dtype=np.int32); n = C.c_int(0); kc = C.c_int(0)
        c.check(c.lib.slam_kpset_download_first(c.h, self.h, s, L.ptr(f), L.ptr(k, L.i32p), self.cap, C.byref(n), C.byref(kc)))
        return f[:n.value].copy(), k[:n.value].copy(), kc.value

    def triangulate_temporal(self, sp, kf_cw, Twc, kf_cur, kf_lo=None, max_error=3.0, min_depth=0.1, min_parallax=20.0, n_bound=0, ctx=None):
        """triangulate_temporal! (mapper.jl:185-262) on the lists (slam_kpset_triangulate_temporal).  kf_cw: (S, nkf, 4, 4) world ->
        camera poses of the key-frames, key-frame id k of stream s at [s, k % nkf]; Twc: (S, 4, 4) camera -> world of the frame;
        kf_cur: (S,) the frame's key-frame id; kf_lo: (S,) oldest id still in the table (default kf_cur - nkf + 1).  The per-observer
        matrices of mapper.jl:226-231 are formed here (numpy), as the Julia caller would form them."""
        c = ctx or self.ctx
        sp = np.ascontiguousarray(sp, dtype=np.float64).reshape(self.S, 32)
        kf_cw = np.asarray(kf_cw, dtype=np.float64).reshape(self.S, -1, 4, 4)
        nkf = kf_cw.shape[1]
        Twc = np.broadcast_to(np.asarray(Twc, dtype=np.float64).reshape(-1, 4, 4), (self.S, 4, 4))
        kf_cur = np.ascontiguousarray(np.broadcast_to(np.asarray(kf_cur, dtype=np.int32), (self.S,)))
        kf_lo = np.ascontiguousarray(np.maximum(kf_cur - nkf + 1, 0) if kf_lo is None else np.broadcast_to(np.asarray(kf_lo, dtype=np.int32), (self.S,)), dtype=np.int32)
        tab = np.zeros((self.S, nkf, 4, 16))
        for s in range(self.S):
            fx, fy, cx, cy = sp[s, 16:20]
            K4 = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
            for k in range(nkf):
                rel = kf_cw[s, k] @ Twc[s]                                # observer_kf.cw * frame.wc
                rel_inv = np.linalg.inv(rel)
                for j, M in enumerate((K4 @ rel_inv, rel_inv, rel, np.linalg.inv(kf_cw[s, k]))):
                    tab[s, k, j] = M.T.reshape(16)                        # column-major
        tab = np.ascontiguousarray(tab.reshape(self.S, nkf, 64))
        c.check(c.lib.slam_kpset_triangulate_temporal(c.h, self.h, L.ptr(sp), L.ptr(tab), int(nkf), L.ptr(kf_cur, L.i32p), L.ptr(kf_lo, L.i32p),
                                                      float(max_error), float(min_depth), float(min_parallax), int(n_bound)))

    def compute_pose_5pt(self, sp, min_parallax=5.0, max_repr_error=3.0, iters=128, seed=0, ctx=None, fetch=True):
        """compute_pose_5pt! (front_end.jl:242-332) for every stream on the device-resident lists (slam_kpset_compute_pose_5pt):
        returns (Rt (S, 3, 4) key-frame -> frame with |t| = 1, status (S,), inlier counts (S,), average parallax (S,), list lengths
        after the removals (S,)).  sp: stream_params(...) with R_compensation in the rotation part of the Tcw slot (rows / columns
        0..2) and the camera / distortion entries filled."""
        c = ctx or self.ctx
        sp = np.ascontiguousarray(sp, dtype=np.float64).reshape(self.S, 32).copy()
        # the seam reads R_compensation as a dense column-major 3 x 3 at [0..8]: repack from the 4 x 4 slot (column-major, stride 4)
        T = sp[:, :16].reshape(self.S, 4, 4)                                       # [col][row]
        sp[:, :9] = T[:, :3, :3].reshape(self.S, 9)
        if not fetch:                                            # enqueue only: the filter acts on the lists, nothing comes back
            c.check(c.lib.slam_kpset_compute_pose_5pt(c.h, self.h, L.ptr(sp), float(min_parallax), float(max_repr_error), int(iters),
                                                      int(seed) & 0xFFFFFFFFFFFFFFFF, None, None, None, None, None))
            return None
        P = np.zeros((self.S, 12)); status = np.zeros(self.S, dtype=np.int32); ninl = np.zeros(self.S, dtype=np.int32)
        par = np.zeros(self.S); counts = np.zeros(self.S, dtype=np.int32)
        c.check(c.lib.slam_kpset_compute_pose_5pt(c.h, self.h, L.ptr(sp), float(min_parallax), float(max_repr_error), int(iters),
                                                  int(seed) & 0xFFFFFFFFFFFFFFFF, L.ptr(P), L.ptr(status, L.i32p), L.ptr(ninl, L.i32p),
                                                  L.ptr(par), L.ptr(counts, L.i32p)))
        return P.reshape(self.S, 4, 3).transpose(0, 2, 1).copy(), status, ninl, par, counts

    def compute_pose(self, sp, threshold=3.0, iters=256, seed=0, pnp_iters_fast=5, pnp_iterations=10, depth_eps=1e-6, repr_eps=None, ctx=None):
        """compute_pose! (front_end.jl:132-219) for every stream on the device-resident lists (slam_kpset_compute_pose): returns
        (poses (S, 4, 4) world -> camera, status (S,), P3P inlier counts (S,), list lengths after the outlier removals (S,)).
        sp: stream_params(...) with the camera / distortion entries filled; repr_eps defaults to `threshold`
        (max_reprojection_error on both, front_end.jl:166,206)."""
        c = ctx or self.ctx
        sp = np.ascontiguousarray(sp, dtype=np.float64).reshape(self.S, 32)
        poses = np.zeros((self.S, 16)); status = np.zeros(self.S, dtype=np.int32); ninl = np.zeros(self.S, dtype=np.int32)
        counts = np.zeros(self.S, dtype=np.int32)
        c.check(c.lib.slam_kpset_compute_pose(c.h, self.h, L.ptr(sp), float(threshold), int(iters), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                              int(pnp_iters_fast), int(pnp_iterations), float(depth_eps),
                                              float(threshold if repr_eps is None else repr_eps),
                                              L.ptr(poses), L.ptr(status, L.i32p), L.ptr(ninl, L.i32p), L.ptr(counts, L.i32p)))
        return poses.reshape(self.S, 4, 4).transpose(0, 2, 1).copy(), status, ninl, counts


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def pose_samples(seed, stream, n, iters):
    """The triples slam_kpset_compute_pose draws for stream `stream` with `n` 3-D keypoints (csrc/pose.hip, k_kpose_samples):
    three distinct indices per iteration, splitmix64(seed ^ stream << 48 ^ iteration << 16 ^ attempt) mod n; -1 when n < 5."""
    out = np.full((iters, 3), -1, dtype=np.int32)
    if n < 5:
        return out
    for it in range(iters):
        att = 0
        idx = []
        while len(idx) < 3:
            h = _splitmix64((int(seed) ^ (int(stream) << 48) ^ (it << 16) ^ att) & 0xFFFFFFFFFFFFFFFF)
            att += 1
            c = int(h % n)
            if c not in idx:
                idx.append(c)
        out[it] = idx
    return out


def pose_samples5(seed, stream, n, iters):
    """The 5-tuples slam_kpset_compute_pose_5pt draws (csrc/fivepoint.hip, k_kfive_samples): as pose_samples, five distinct indices."""
    out = np.full((iters, 5), -1, dtype=np.int32)
    for it in range(iters):
        att = 0
        idx = []
        while len(idx) < 5:
            h = _splitmix64((int(seed) ^ (int(stream) << 48) ^ (it << 16) ^ att) & 0xFFFFFFFFFFFFFFFF)
            att += 1
            c = int(h % n)
            if c not in idx:
                idx.append(c)
        out[it] = idx
    return out


def pose_5pt_inputs(cam, dist, yx, kyx):
    """front_end.jl:263-272 for the keypoints both frames observe: (px1, px2, pd1, pd2), (x, y) order -- undistorted pixels and
    normalised coordinates of the key-frame (1) and the frame (2); the arithmetic of k_kfive_gather."""
    fx, fy, cx, cy = cam
    z = np.zeros((len(np.atleast_2d(yx)), 3))
    _, p2, _ = pose_inputs(cam, dist, yx, z)
    _, p1, _ = pose_inputs(cam, dist, kyx, z)
    pd = lambda p: np.stack([(p[:, 0] - cx) / fx, (p[:, 1] - cy) / fy], axis=1)
    return p1, p2, pd(p1), pd(p2)


def pose_5pt_compose(Rt, prev_cw, cur_wc):
    """front_end.jl:320-330: translation scaled to the motion model's key-frame -> frame distance, composed with the key-frame pose."""
    scale = np.linalg.norm((prev_cw @ cur_wc)[:3, 3])
    R, t = Rt[:, :3], Rt[:, 3]
    t = scale * (t / np.linalg.norm(t))
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T @ prev_cw


def pose_inputs(cam, dist, yx, xyz):
    """front_end.jl:139-160 for one stream's 3-D keypoints: (pts3d, px_xy, pdn) as p3p_ransac takes them -- undistort_point
    (camera.jl:98-125), backproject (:138-140), normalize; the arithmetic of k_kpose_gather term by term."""
    fx, fy, cx, cy = cam
    k1, k2, p1, p2 = dist
    yx = np.asarray(yx, dtype=np.float64).reshape(-1, 2)
    ny = (yx[:, 0] - cy) / fy; nx = (yx[:, 1] - cx) / fx
    s0 = ny * ny; s1 = nx * nx; r2 = s0 + s1
    rd = (1.0 + k1 * r2) + k2 * (r2 * r2)
    pp = ny * nx
    dtx = 2 * p1 * pp + p2 * (r2 + 2 * s0); dty = p1 * (r2 + 2 * s1) + 2 * p2 * pp
    uy = (rd * ny + dty) * fy + cy; ux = (rd * nx + dtx) * fx + cx
    bx = (ux - cx) / fx; by = (uy - cy) / fy
    inv = 1.0 / np.sqrt((bx * bx + by * by) + 1.0)
    px_xy = np.stack([ux, uy], axis=1)
    pdn = np.stack([inv * bx, inv * by, inv * 1.0], axis=1)
    return np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3), px_xy, pdn
