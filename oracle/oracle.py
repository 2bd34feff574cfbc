"""ctypes front-end of the CPU oracle (oracle/libslam_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/slam_oracle.h.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product
package.  Parity with the Julia reference is UNPINNED (argued from source, not
measured): Julia is not installed and the reference ships no golden vectors.

Array conventions follow the reference: images are H x W Float64 in Julia's
column-major layout, which in numpy is ``np.asfortranarray(img)``; points are
``(n, 2)`` C-contiguous ``(y, x)`` 1-based Float64.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    so = os.path.join(_HERE, "libslam_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_image.c", "orc_lk.c", "orc_ba.c", "orc_tri.c", "orc_p3p.c", "orc_5pt.c", "slam_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libslam_oracle.so"])
    return so


def use_native():
    """Switch this process to an -O3 -march=native build of the same sources, compiled here into a temporary directory (bench.py's
    cpu_baseline leg; a host-specific binary must not travel with the tree).  Returns the flags used, or None if it did not build."""
    global _LIB
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"libslam_oracle_native_{os.getuid()}.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native", f"NATIVE_OUT={out}"])
        l = C.CDLL(out)
    except (subprocess.CalledProcessError, OSError):
        return None
    l.orc_bilinear.restype = C.c_double
    l.orc_bilinear.argtypes = [f64p, C.c_int, C.c_int, C.c_double, C.c_double]
    l.orc_pyr_layout.restype = C.c_int64
    _LIB = l
    return "-O3 -march=native -ffp-contract=off -fno-fast-math"


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_bilinear.restype = C.c_double
        _LIB.orc_bilinear.argtypes = [f64p, C.c_int, C.c_int, C.c_double, C.c_double]
        _LIB.orc_pyr_layout.restype = C.c_int64
    return _LIB


def _p(a, t=f64p):
    return a.ctypes.data_as(t)


def fimg(img):
    """H x W array -> Fortran-ordered float64 (Julia memory layout)."""
    return np.asfortranarray(img, dtype=np.float64)


# ----------------------------------------------------------------------------
def gaussian_taps(sigma):
    w = np.zeros(4 * int(np.ceil(sigma)) + 1)
    n = lib().orc_gaussian_taps(C.c_double(sigma), _p(w))
    return w[:n]


def imfilter_sep(img, k1, k2, border=0):
    img = fimg(img)
    H, W = img.shape
    out = np.empty_like(img, order="F")
    k1 = np.ascontiguousarray(k1, dtype=np.float64)
    k2 = np.ascontiguousarray(k2, dtype=np.float64)
    lib().orc_imfilter_sep(_p(out), _p(img), H, W, _p(k1), len(k1), _p(k2), len(k2), border)
    return out


def iir_gaussian(img, sigma, border=0):
    img = fimg(img)
    H, W = img.shape
    out = np.empty_like(img, order="F")
    lib().orc_iir_gaussian(_p(out), _p(img), H, W, C.c_double(sigma), border)
    return out


def iir_coeffs(sigma):
    a = np.zeros(3); scale = C.c_double(); M = np.zeros(9); asum = C.c_double()
    lib().orc_iir_coeffs(C.c_double(sigma), _p(a), C.byref(scale), _p(M), C.byref(asum))
    return a, scale.value, M.reshape(3, 3), asum.value


def imresize(img, Hd, Wd):
    img = fimg(img)
    H, W = img.shape
    out = np.empty((Hd, Wd), order="F")
    lib().orc_imresize(_p(out), Hd, Wd, _p(img), H, W)
    return out


def bilinear(img, r, c):
    img = fimg(img)
    return lib().orc_bilinear(_p(img), img.shape[0], img.shape[1], C.c_double(r), C.c_double(c))


def get_mask(H, W, pts_yx, radius):
    pts = np.ascontiguousarray(pts_yx, dtype=np.float64).reshape(-1, 2)
    m = np.empty((H, W), order="F")
    lib().orc_get_mask(_p(m), H, W, _p(pts), len(pts), radius)
    return m


def shi_tomasi(cell):
    cell = fimg(cell)
    h, w = cell.shape
    out = np.empty((h, w), order="F")
    lib().orc_shi_tomasi(_p(out), _p(cell), h, w, h)
    return out


def grid_resolution(H, W, cell_size):
    """SlamManager: ceil.(Int, (H, W) ./ max_distance), src/SLAM.jl:150-151."""
    return -(-H // cell_size), -(-W // cell_size)


def detect(img, cur_yx, max_points=1000, radius=17, cell_size=35, sigma_mask=3.0, min_response=1e-4, grid=None):
    img = fimg(img)
    H, W = img.shape
    gr, gc = grid if grid is not None else grid_resolution(H, W, cell_size)
    cur = np.ascontiguousarray(cur_yx, dtype=np.float64).reshape(-1, 2)
    n_cur = len(cur)
    k = max(1, int(np.ceil(max(max_points - n_cur, 0) / (gr * gc))))
    cap = gr * gc * k + 8
    out = np.zeros((cap, 2), dtype=np.int64)
    n = lib().orc_detect(_p(img), H, W, _p(cur), n_cur, max_points, radius, gr, gc, cell_size,
                         C.c_double(sigma_mask), C.c_double(min_response), _p(out, i64p), cap)
    assert n >= 0
    return out[:n].copy()


def describe(img, rc, pattern, sigma=np.sqrt(2.0), window=9):
    img = fimg(img)
    H, W = img.shape
    rc = np.ascontiguousarray(rc, dtype=np.int64).reshape(-1, 2)
    pattern = np.ascontiguousarray(pattern, dtype=np.int32).reshape(-1, 4)
    nb = len(pattern)
    bits = np.zeros((len(rc), nb // 64), dtype=np.uint64)
    orc = np.zeros((len(rc), 2), dtype=np.int64)
    n = lib().orc_describe(_p(img), H, W, _p(rc, i64p), len(rc), _p(pattern, i32p), nb,
                           C.c_double(sigma), window, _p(bits, u64p), _p(orc, i64p))
    return bits[:n].copy(), orc[:n].copy()


# ----------------------------------------------------------------------------
class Pyramid:
    """Host mirror of LKPyramid (pyramid.jl:16-24) as six flat plane buffers."""
    PLANES = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")

    def __init__(self, H, W, total_levels):
        self.H0, self.W0, self.levels = H, W, total_levels
        Hs = (C.c_int * 8)(); Ws = (C.c_int * 8)(); off = (C.c_int64 * 9)()
        self.total = lib().orc_pyr_layout(H, W, total_levels, Hs, Ws, off)
        self.Hs, self.Ws, self.off = list(Hs)[:total_levels], list(Ws)[:total_levels], list(off)[:total_levels + 1]
        for n in self.PLANES:
            setattr(self, n, np.zeros(self.total))

    def plane(self, name, level):
        """level is 0-based here; returns an H x W Fortran-ordered view."""
        buf = getattr(self, name)
        return buf[self.off[level]:self.off[level + 1]].reshape((self.Hs[level], self.Ws[level]), order="F")


def pyr_build(img, pyramid_levels=3, sigma=1.0, mode=1):
    img = fimg(img)
    H, W = img.shape
    p = Pyramid(H, W, pyramid_levels + 1)
    lib().orc_pyr_build_flat(_p(img), H, W, p.levels, C.c_double(sigma), mode,
                             _p(p.layers), _p(p.Iy), _p(p.Ix), _p(p.Iyy), _p(p.Ixx), _p(p.Iyx))
    return p


def fb_tracking(prev, cur, pts_yx, disp0=None, iterations=30, window=9, pyramid_levels=3,
                eig_thr=1e-4, eps=1e-2, max_distance=1.0, sum_order=0, threads=1):
    pts = np.ascontiguousarray(pts_yx, dtype=np.float64).reshape(-1, 2)
    n = len(pts)
    out = np.full((n, 2), np.nan)
    status = np.zeros(n, dtype=np.uint8)
    d0 = None if disp0 is None else np.ascontiguousarray(disp0, dtype=np.float64).reshape(-1, 2)
    rc = lib().orc_fb_tracking_flat(
        prev.H0, prev.W0, prev.levels,
        _p(prev.layers), _p(prev.Iy), _p(prev.Ix), _p(prev.Iyy), _p(prev.Ixx), _p(prev.Iyx),
        _p(cur.layers), _p(cur.Iy), _p(cur.Ix), _p(cur.Iyy), _p(cur.Ixx), _p(cur.Iyx),
        _p(pts), None if d0 is None else _p(d0), n, iterations, window, pyramid_levels,
        C.c_double(eig_thr), C.c_double(eps), C.c_double(max_distance), _p(out), _p(status, u8p),
        sum_order, threads)
    if rc != 0:
        raise RuntimeError("Not enough layers in pyramids.")
    return out, status.astype(bool)


def svd2x2(M):
    M = np.asfortranarray(M, dtype=np.float64)
    U = np.zeros((2, 2), order="F"); S = np.zeros(2); V = np.zeros((2, 2), order="F")
    lib().orc_svd2x2(_p(M), _p(U), _p(S), _p(V))
    return U, S, V


def pinv2x2(M):
    M = np.asfortranarray(M, dtype=np.float64)
    G = np.zeros((2, 2), order="F"); S = np.zeros(2)
    lib().orc_pinv2x2(_p(M), _p(G), _p(S))
    return G, S


# ----------------------------------------------------------------------------
def rotzyx(t1, t2, t3):
    R = np.zeros(9)
    lib().orc_rotzyx(C.c_double(t1), C.c_double(t2), C.c_double(t3), _p(R))
    return R.reshape(3, 3)


def rotzyx_angles(R):
    R = np.ascontiguousarray(R, dtype=np.float64)
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().orc_rotzyx_angles(_p(R), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def bundle_adjustment(cam, theta, theta_const, pixels_yx, pose_ids, point_ids,
                      iters_fast=5, iterations=10, repr_eps=5.0, solver=1):
    """Returns (theta_new, outliers, stats dict).  cam = (fx, fy, cx, cy)."""
    theta = np.array(theta, dtype=np.float64, copy=True)
    tc = np.ascontiguousarray(theta_const, dtype=np.uint8)
    px = np.ascontiguousarray(pixels_yx, dtype=np.float64).reshape(-1, 2)
    pi = np.ascontiguousarray(pose_ids, dtype=np.int64)
    li = np.ascontiguousarray(point_ids, dtype=np.int64)
    P, O = len(tc), len(pi)
    M = (len(theta) - 6 * P) // 3
    outl = np.zeros(O, dtype=np.uint8)
    st = np.zeros(8)
    lib().orc_bundle_adjustment_flat(C.c_double(cam[0]), C.c_double(cam[1]), C.c_double(cam[2]), C.c_double(cam[3]),
                                     P, M, O, _p(theta), _p(tc, u8p), _p(px), _p(pi, i64p), _p(li, i64p),
                                     _p(outl, u8p), iters_fast, iterations, C.c_double(repr_eps), solver, _p(st))
    stats = dict(ssr_init=st[0], ssr_pass1=st[1], ssr_final=st[2], iters_pass1=int(st[3]), iters_pass2=int(st[4]),
                 n_outliers=int(st[5]), inner_iters=int(st[6]))
    return theta, outl.astype(bool), stats


class _BAProblem(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("P", C.c_int), ("M", C.c_int), ("O", C.c_int),
                ("theta", f64p), ("theta_const", u8p), ("pixels_yx", f64p),
                ("pose_ids", i64p), ("point_ids", i64p), ("outliers", u8p)]


def ba_reduced_system(cam, theta, theta_const, pixels_yx, pose_ids, point_ids, outliers, ignore_outliers,
                      inv_delta, m_begin, m_end):
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    tc = np.ascontiguousarray(theta_const, dtype=np.uint8)
    px = np.ascontiguousarray(pixels_yx, dtype=np.float64).reshape(-1, 2)
    pi = np.ascontiguousarray(pose_ids, dtype=np.int64)
    li = np.ascontiguousarray(point_ids, dtype=np.int64)
    ol = np.ascontiguousarray(outliers, dtype=np.uint8)
    P, O = len(tc), len(pi)
    M = (len(theta) - 6 * P) // 3
    prob = _BAProblem(cam[0], cam[1], cam[2], cam[3], P, M, O, _p(theta), _p(tc, u8p), _p(px), _p(pi, i64p),
                      _p(li, i64p), _p(ol, u8p))
    n = 6 * P
    S = np.zeros((n, n), order="F"); g2 = np.zeros(2 * n); ssr = C.c_double()
    lib().orc_ba_reduced_system(C.byref(prob), _p(theta), int(ignore_outliers), C.c_double(inv_delta),
                                m_begin, m_end, _p(S), _p(g2), C.byref(ssr))
    return S, g2[:n], g2[n:], ssr.value


def ba_residuals(cam, theta, theta_const, pixels_yx, pose_ids, point_ids):
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    tc = np.ascontiguousarray(theta_const, dtype=np.uint8)
    px = np.ascontiguousarray(pixels_yx, dtype=np.float64).reshape(-1, 2)
    pi = np.ascontiguousarray(pose_ids, dtype=np.int64)
    li = np.ascontiguousarray(point_ids, dtype=np.int64)
    P, O = len(tc), len(pi)
    M = (len(theta) - 6 * P) // 3
    ol = np.zeros(O, dtype=np.uint8)
    prob = _BAProblem(cam[0], cam[1], cam[2], cam[3], P, M, O, _p(theta), _p(tc, u8p), _p(px), _p(pi, i64p),
                      _p(li, i64p), _p(ol, u8p))
    Y = np.zeros(2 * O)
    lib().orc_ba_residuals(C.byref(prob), _p(theta), 0, _p(Y))
    return Y


def pnp_ba(cam, pose_cw, pixels_yx, points_xyz, iters_fast=5, iterations=10, depth_eps=1e-6, repr_eps=5.0):
    pose = np.asfortranarray(pose_cw, dtype=np.float64)
    px = np.ascontiguousarray(pixels_yx, dtype=np.float64).reshape(-1, 2)
    pts = np.ascontiguousarray(points_xyz, dtype=np.float64).reshape(-1, 3)
    n = len(px)
    out = np.zeros((4, 4), order="F")
    e0, e1, no = C.c_double(), C.c_double(), C.c_int()
    outl = np.zeros(n, dtype=np.uint8)
    lib().orc_pnp_ba(C.c_double(cam[0]), C.c_double(cam[1]), C.c_double(cam[2]), C.c_double(cam[3]), _p(pose),
                     _p(px), _p(pts), n, iters_fast, iterations, C.c_double(depth_eps), C.c_double(repr_eps),
                     _p(out), C.byref(e0), C.byref(e1), _p(outl, u8p), C.byref(no))
    return np.array(out), e0.value, e1.value, outl.astype(bool), no.value


def triangulate(P1, P2, T21, cam1, cam2, px1_yx, px2_yx, max_error, min_depth=0.1, parallax=None, min_parallax=20.0):
    """Array-level body of triangulate_stereo! (parallax=None) / triangulate_temporal! (mapper.jl:142-262).
    P1, P2, T21: 4x4; cam = (fx, fy, cx, cy); pixels (n, 2) (y, x).  Returns (xyz (n, 3) in camera-1 coordinates, status)."""
    P1 = np.asfortranarray(P1, dtype=np.float64); P2 = np.asfortranarray(P2, dtype=np.float64); T = np.asfortranarray(T21, dtype=np.float64)
    c1 = np.ascontiguousarray(cam1, dtype=np.float64); c2 = np.ascontiguousarray(cam2, dtype=np.float64)
    a = np.ascontiguousarray(px1_yx, dtype=np.float64).reshape(-1, 2); b = np.ascontiguousarray(px2_yx, dtype=np.float64).reshape(-1, 2)
    n = len(a)
    out = np.zeros((n, 3)); st = np.zeros(n, dtype=np.uint8)
    par = None if parallax is None else np.ascontiguousarray(parallax, dtype=np.float64)
    lib().orc_triangulate(_p(P1), _p(P2), _p(T), _p(c1), _p(c2), _p(a), _p(b), n, C.c_double(max_error), C.c_double(min_depth),
                          _p(par) if par is not None else None, C.c_double(min_parallax), _p(out), _p(st, u8p))
    return out, st.astype(bool)


def sym4_min_eigvec(S, inverse_iteration=False):
    S = np.array(S, dtype=np.float64, order="C").copy()
    v = np.zeros(4)
    (lib().orc_sym4_min_eigvec_invit if inverse_iteration else lib().orc_sym4_min_eigvec)(_p(S), _p(v))
    return v


def quartic_real_roots(A):
    """Real roots of A[4] x^4 + ... + A[0] (orc_p3p.c)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    r = np.zeros(4)
    n = lib().orc_quartic_real_roots(_p(A), _p(r))
    return r[:n]


def p3p_solve(X, F):
    """Minimal solver: X 3x3 world points (rows), F 3x3 bearing vectors (rows) -> list of 3x4 [R | t]."""
    X = np.ascontiguousarray(X, dtype=np.float64); F = np.ascontiguousarray(F, dtype=np.float64)
    Rt = np.zeros(48)
    ns = lib().orc_p3p_solve(_p(X), _p(F), _p(Rt))
    return [Rt[12 * s:12 * s + 12].reshape(4, 3).T.copy() for s in range(ns)]


def p3p_ransac(pts3d, px_xy, pdn, K, threshold, samples):
    """p3p_ransac of compute_pose! (front_end.jl:164-167) over caller-supplied 0-based sample triples.
    Returns (n_inliers, KP 3x4, Rt 3x4, inliers bool, error, best_iter)."""
    pts = np.ascontiguousarray(pts3d, dtype=np.float64).reshape(-1, 3)
    px = np.ascontiguousarray(px_xy, dtype=np.float64).reshape(-1, 2)
    bd = np.ascontiguousarray(pdn, dtype=np.float64).reshape(-1, 3)
    Kf = np.asfortranarray(K, dtype=np.float64)
    sm = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 3)
    n = len(pts)
    KP = np.zeros((3, 4), order="F"); Rt = np.zeros((3, 4), order="F")
    inl = np.zeros(n, dtype=np.uint8); err = C.c_double(); bi = C.c_int()
    cnt = lib().orc_p3p_ransac(_p(pts), _p(px), _p(bd), n, _p(Kf), C.c_double(threshold), _p(sm, i32p), len(sm),
                               _p(KP), _p(Rt), _p(inl, u8p), C.byref(err), C.byref(bi))
    return cnt, np.array(KP), np.array(Rt), inl.astype(bool), err.value, bi.value


def poly_real_roots(p):
    """Ascending real roots of p[0] + p[1] x + ... (degree <= 10; orc_5pt.c)."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    r = np.zeros(10)
    n = lib().orc_poly_real_roots(_p(p), len(p) - 1, _p(r))
    return r[:n]


def five_point_solve(q1, q2):
    """Nister's minimal solver: q1, q2 (5, 2) normalised (x, y) with q2' E q1 = 0 -> list of 3x3 E."""
    a = np.ascontiguousarray(q1, dtype=np.float64).reshape(5, 2); b = np.ascontiguousarray(q2, dtype=np.float64).reshape(5, 2)
    Es = np.zeros(90)
    ne = lib().orc_five_point_solve(_p(a), _p(b), _p(Es))
    return [Es[9 * i:9 * i + 9].reshape(3, 3).copy() for i in range(ne)]


def essential_poses(E):
    """The four [R | t] (3x4) of an essential matrix."""
    E = np.ascontiguousarray(E, dtype=np.float64)
    Rt = np.zeros(48)
    k = lib().orc_essential_poses(_p(E), _p(Rt))
    return [Rt[12 * i:12 * i + 12].reshape(4, 3).T.copy() for i in range(k)]


def five_point_ransac(px1_xy, px2_xy, pd1_xy, pd2_xy, K1, K2, max_repr_error, samples):
    """five_point_ransac of compute_pose_5pt! (front_end.jl:305-308) over caller-supplied 0-based 5-tuples.
    Returns (n_inliers, E 3x3, P 3x4, inliers bool, error, best_iter)."""
    a = np.ascontiguousarray(px1_xy, dtype=np.float64).reshape(-1, 2); b = np.ascontiguousarray(px2_xy, dtype=np.float64).reshape(-1, 2)
    c = np.ascontiguousarray(pd1_xy, dtype=np.float64).reshape(-1, 2); d = np.ascontiguousarray(pd2_xy, dtype=np.float64).reshape(-1, 2)
    k1 = np.asfortranarray(K1, dtype=np.float64); k2 = np.asfortranarray(K2, dtype=np.float64)
    sm = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 5)
    n = len(a)
    E = np.zeros((3, 3), order="F"); P = np.zeros((3, 4), order="F")
    inl = np.zeros(max(n, 1), dtype=np.uint8); err = C.c_double(); bi = C.c_int()
    cnt = lib().orc_five_point_ransac(_p(a), _p(b), _p(c), _p(d), n, _p(k1), _p(k2), C.c_double(max_repr_error), _p(sm, i32p), len(sm),
                                      _p(E), _p(P), _p(inl, u8p), C.byref(err), C.byref(bi))
    return cnt, np.array(E), np.array(P), inl[:n].astype(bool), err.value, bi.value


# ----------------------------------------------------------------------------
def undistort_point(cam, dist, p_yx):
    """undistort_point / undistort_pdn_point, src/camera.jl:98-125 (applies the lens model to a (y, x) pixel).
    cam = (fx, fy, cx, cy); dist = (k1, k2, p1, p2)."""
    fx, fy, cx, cy = cam
    k1, k2, p1, p2 = dist
    ny = (p_yx[0] - cy) / fy; nx = (p_yx[1] - cx) / fx                     # :99-101
    s0 = ny * ny; s1 = nx * nx
    r2 = s0 + s1                                                           # :114
    rd = 1.0 + k1 * r2 + k2 * r2 ** 2                                      # :116
    p = ny * nx                                                            # :118
    dtx = 2 * p1 * p + p2 * (r2 + 2 * s0)                                  # :119
    dty = p1 * (r2 + 2 * s1) + 2 * p2 * p                                  # :120
    dy = rd * ny + dty; dx = rd * nx + dtx                                 # :122
    return np.array([dy * fy + cy, dx * fx + cx])                          # :124


def optical_flow_matching(prev, cur, pixels, is_3d, projections, image_size, stereo=False,
                          undistorted_left=None, right_cam=None, right_dist=(0.0, 0.0, 0.0, 0.0),
                          pyramid_levels=3, window_size=9, max_distance=1.0, epipolar_error=2.0,
                          sum_order=0, threads=1):
    """Array-level restatement of optical_flow_matching!(map_manager, frame, from, to, stereo)
    -- src/map_manager.jl:451-564 with maybe_stereo_update! :579-590.

    Keypoint j has `pixels[j]` (y, x), flag `is_3d[j]`, and for 3-D keypoints the projection of its map point into
    the target image `projections[j]` (project_world_to_image_distort / ..._right_image_distort, :485-488).
    image_size = (height, width) of in_image / in_right_image (camera.jl:90-92).

    Returns a dict of arrays over the input keypoints:
      new_pixels  (n, 2)  position written by update_keypoint! / update_stereo_keypoint!  (input pixel elsewhere)
      updated     (n,)    keypoint was updated  (:526-531, :554-558)
      removed     (n,)    observation removed   (:496-497 stereo out-of-image, :559 failed 2-D track of a temporal match)
    A 3-D keypoint whose projection is outside the image is skipped entirely in the temporal case (:501-506:
    neither tracked, updated nor removed) and removed in the stereo case (:491-498).
    """
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    n = len(px)
    is3 = np.asarray(is_3d).astype(bool)
    proj = np.ascontiguousarray(projections, dtype=np.float64).reshape(-1, 2)
    Himg, Wimg = image_size if image_size is not None else (np.inf, np.inf)     # None: every projection counts as inside
    new = px.copy(); updated = np.zeros(n, bool); removed = np.zeros(n, bool)
    pyramid_levels_3d = 1                                                   # :458
    scale = 1.0 / 2.0 ** pyramid_levels_3d                                  # :466
    ids, ids3d, disp3d = [], [], []
    for j in range(n):                                                      # :471-508
        if not is3[j]:
            ids.append(j); continue
        inside = image_size is None or ((1 <= proj[j, 0] <= Himg) and (1 <= proj[j, 1] <= Wimg))   # camera.jl:91
        if inside:
            ids3d.append(j); disp3d.append(scale * (proj[j] - px[j]))       # :494 / :504
        elif stereo:
            removed[j] = True                                               # :496-497

    def stereo_update(j, new_pos):                                          # maybe_stereo_update! :579-590
        right_pixel = undistort_point(right_cam, right_dist, new_pos)
        if abs(undistorted_left[j, 0] - right_pixel[0]) > epipolar_error:
            return False
        new[j] = (px[j, 0], new_pos[1]); updated[j] = True                  # :587-588
        return True

    if ids3d:                                                               # :516-540
        nk, st = fb_tracking(prev, cur, px[ids3d], disp0=np.array(disp3d), pyramid_levels=pyramid_levels_3d,
                             window=window_size, max_distance=max_distance, sum_order=sum_order, threads=threads)
        for k, j in enumerate(ids3d):
            if st[k]:
                if stereo:
                    stereo_update(j, nk[k])
                else:
                    new[j] = nk[k]; updated[j] = True                       # :530
            else:
                ids.append(j)                                               # :533-537: re-tracked with the 2-D set, no prior
    if ids:                                                                 # :546-562
        nk, st = fb_tracking(prev, cur, px[ids], pyramid_levels=pyramid_levels, window=window_size,
                             max_distance=max_distance, sum_order=sum_order, threads=threads)
        for k, j in enumerate(ids):
            if stereo:
                if st[k]:
                    stereo_update(j, nk[k])
            elif st[k]:
                new[j] = nk[k]; updated[j] = True                           # :558
            else:
                removed[j] = True                                           # :559
    return dict(new_pixels=new, updated=updated, removed=removed)
