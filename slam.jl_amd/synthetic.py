"""Seeded synthetic workloads of the shapes BASELINE.json names (there is no
network for KITTI/EuRoC): textured stereo streams with known motion for the
front-end, and windowed bundle-adjustment scenes.  Host-side numpy only."""
import numpy as np

SHAPES = {
    "kitti05": (370, 1226),   # true KITTI 04-12 size
    "kitti00": (376, 1241),   # as hard-coded in example/kitty/main.jl:21
    "euroc": (480, 640),
    "fhd": (1080, 1920),
}
KITTI_CAM = (718.856, 718.856, 607.1928, 185.2157)  # fx, fy, cx, cy of KITTI 00-02/05-ish


def texture_canvas(H, W, seed=0, margin=64):
    """Band-limited random texture in [0,1], (H+2m) x (W+2m)."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(0x51A7 + seed)
    Hc, Wc = H + 2 * margin, W + 2 * margin
    fine = gaussian_filter(rng.standard_normal((Hc, Wc)), 2.0)
    mid = gaussian_filter(rng.standard_normal((Hc, Wc)), 6.0)
    coarse = gaussian_filter(rng.standard_normal((Hc, Wc)), 20.0)
    t = fine / fine.std() + 0.7 * mid / mid.std() + 0.5 * coarse / coarse.std()
    t = (t - t.min()) / (t.max() - t.min())
    return t


def render(canvas, H, W, dy, dx, margin=64):
    """Image whose content is the canvas translated by (dy, dx) px (sub-pixel, cubic):
    a scene point at image position p in the (0,0) render is at p + (dy, dx)."""
    from scipy.ndimage import shift
    s = shift(canvas, (dy, dx), order=3, mode="nearest")
    return np.asfortranarray(np.clip(s[margin:margin + H, margin:margin + W], 0.0, 1.0))


def stereo_stream(shape="kitti05", n_frames=8, seed=0, step=(1.3, -2.1), disparity=12.4):
    """Returns (left[n], right[n], flows[n]) with flows[i] = cumulative (dy,dx) of frame i."""
    H, W = SHAPES[shape] if isinstance(shape, str) else shape
    canvas = texture_canvas(H, W, seed)
    left, right, flows = [], [], []
    for i in range(n_frames):
        dy, dx = step[0] * i, step[1] * i
        left.append(render(canvas, H, W, dy, dx))
        right.append(render(canvas, H, W, dy, dx - disparity))
        flows.append((dy, dx))
    return left, right, flows


def rotzyx(t1, t2, t3):
    s1, c1, s2, c2, s3, c3 = np.sin(t1), np.cos(t1), np.sin(t2), np.cos(t2), np.sin(t3), np.cos(t3)
    return np.array([[c1 * c2, c1 * s2 * s3 - s1 * c3, c1 * s2 * c3 + s1 * s3],
                     [s1 * c2, s1 * s2 * s3 + c1 * c3, s1 * s2 * c3 - c1 * s3],
                     [-s2, c2 * s3, c2 * c3]])


def _rot_batch(ang):
    """RotZYX for an (n,3) array of angles -> (n,3,3)."""
    s1, c1 = np.sin(ang[:, 0]), np.cos(ang[:, 0]); s2, c2 = np.sin(ang[:, 1]), np.cos(ang[:, 1]); s3, c3 = np.sin(ang[:, 2]), np.cos(ang[:, 2])
    R = np.empty((len(ang), 3, 3))
    R[:, 0, 0] = c1 * c2; R[:, 0, 1] = c1 * s2 * s3 - s1 * c3; R[:, 0, 2] = c1 * s2 * c3 + s1 * s3
    R[:, 1, 0] = s1 * c2; R[:, 1, 1] = s1 * s2 * s3 + c1 * c3; R[:, 1, 2] = s1 * s2 * c3 - c1 * s3
    R[:, 2, 0] = -s2; R[:, 2, 1] = c2 * s3; R[:, 2, 2] = c2 * c3
    return R


def ba_scene(P=5, M=800, obs_per_point=10, seed=0, cam=KITTI_CAM, H=376, W=1241,
             noise_px=0.5, outlier_frac=0.02, n_const=1, perturb=(2e-3, 2e-2, 5e-2)):
    """Windowed BA problem in the reference's flat layout (estimator.jl:16-40):
    theta = [6P (RotZYX t1,t2,t3, t) ; 3M], pixels (O,2) as (y,x), 1-based ids.
    Cameras move forward along +z with a slight curve; each point is seen by a
    contiguous run of min(obs_per_point, P) key-frames; pixels = projection +
    N(0, noise_px); a fraction of observations are wrong associations 4-12 px
    away; the first n_const poses are constant; theta0 = ground truth perturbed
    by N(0, perturb = (rad, m pose, m point)).  Observations are ordered by
    point, then observer (the feeder's order, estimator.jl:186-227)."""
    rng = np.random.default_rng(0xBA00 + seed)
    fx, fy, cx, cy = cam
    k = min(obs_per_point, P)
    idx = np.arange(P, dtype=np.float64)
    ang = np.stack([0.002 * idx, 0.01 * idx, -0.001 * idx], 1)
    Rp = _rot_batch(ang)
    centre = np.stack([1e-4 * idx * idx, np.zeros(P), 0.8 * idx], 1)
    tp = -np.einsum("pij,pj->pi", Rp, centre)
    poses = np.concatenate([ang, tp], 1)
    pts = np.zeros((M, 3)); start = np.zeros(M, dtype=np.int64)
    todo = np.arange(M)
    proj = np.zeros((M, k, 2))
    for _ in range(60):
        if len(todo) == 0:
            break
        n = len(todo)
        st = rng.integers(0, P - k + 1, n)
        mid = st + k // 2
        u = rng.uniform(40, W - 40, n); v = rng.uniform(40, H - 40, n); z = rng.uniform(6.0, 40.0, n) + 0.8 * k
        Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
        Xw = np.einsum("nji,nj->ni", Rp[mid], Xc - tp[mid])            # R^T (Xc - t)
        cams = st[:, None] + np.arange(k)[None, :]                       # (n, k)
        xc = np.einsum("nkij,nj->nki", Rp[cams], Xw) + tp[cams]
        py = fy * xc[..., 1] / xc[..., 2] + cy; pxx = fx * xc[..., 0] / xc[..., 2] + cx
        ok = ((xc[..., 2] > 1.0) & (py >= 1) & (py <= H) & (pxx >= 1) & (pxx <= W)).all(1)
        good = todo[ok]
        pts[good] = Xw[ok]; start[good] = st[ok]
        proj[good, :, 0] = py[ok]; proj[good, :, 1] = pxx[ok]
        todo = todo[~ok]
    assert len(todo) == 0, "could not place all points"
    obs_point = np.repeat(np.arange(1, M + 1, dtype=np.int64), k)
    obs_pose = (start[:, None] + np.arange(k)[None, :] + 1).reshape(-1).astype(np.int64)
    pix = proj.reshape(-1, 2) + rng.normal(0, noise_px, (M * k, 2))
    O = len(pix)
    n_out = int(round(outlier_frac * O))
    out_idx = rng.choice(O, n_out, replace=False) if n_out else np.zeros(0, dtype=int)
    a = rng.uniform(0, 2 * np.pi, n_out); mag = rng.uniform(4.0, 12.0, n_out)
    pix[out_idx, 0] += mag * np.sin(a); pix[out_idx, 1] += mag * np.cos(a)
    theta_gt = np.concatenate([poses.ravel(), pts.ravel()])
    theta0 = theta_gt.copy()
    pp = theta0[:6 * P].reshape(P, 6); lp = theta0[6 * P:].reshape(M, 3)
    const = np.zeros(P, dtype=np.uint8); const[:n_const] = 1
    free = const == 0
    pp[free, :3] += rng.normal(0, perturb[0], (free.sum(), 3))
    pp[free, 3:] += rng.normal(0, perturb[1], (free.sum(), 3))
    lp += rng.normal(0, perturb[2], lp.shape)
    return dict(cam=cam, P=P, M=M, O=O, theta0=theta0, theta_gt=theta_gt, theta_const=const,
                pixels_yx=np.ascontiguousarray(pix), pose_ids=obs_pose, point_ids=obs_point,
                gross_outliers=np.sort(out_idx))


def ba_scene_loop(P=50, M=10000, seed=0, n_loop=1500, k_loop=5, **kw):
    """ba_scene plus `n_loop` loop-closure map points: far points ahead of the track that the first `k_loop` AND the last
    `k_loop` key-frames observe (local_map_matching re-associating old map points, map_manager.jl:300-449).  The reduced camera
    system is no longer block-banded (half-bandwidth P - 1 - n_const): the solver's general path."""
    s = ba_scene(P=P, M=M, seed=seed, **kw)
    rng = np.random.default_rng(0xC105E + seed)
    fx, fy, cx, cy = s["cam"]
    H, W = kw.get("H", 376), kw.get("W", 1241)
    noise_px = kw.get("noise_px", 0.5)
    gt = s["theta_gt"]
    poses = gt[:6 * P].reshape(P, 6)
    Rp = _rot_batch(poses[:, :3]); tp = poses[:, 3:]
    cams = np.concatenate([np.arange(k_loop), np.arange(P - k_loop, P)])
    pts, pix = [], []
    while len(pts) < n_loop:
        u = rng.uniform(200, W - 200); v = rng.uniform(80, H - 80); z = rng.uniform(15.0, 60.0)
        Xc = np.array([(u - cx) / fx * z, (v - cy) / fy * z, z])
        Xw = Rp[P - 1].T @ (Xc - tp[P - 1])
        xc = np.einsum("kij,j->ki", Rp[cams], Xw) + tp[cams]
        py = fy * xc[:, 1] / xc[:, 2] + cy; px = fx * xc[:, 0] / xc[:, 2] + cx
        if (xc[:, 2] > 1.0).all() and (py >= 1).all() and (py <= H).all() and (px >= 1).all() and (px <= W).all():
            pts.append(Xw); pix.append(np.stack([py, px], 1))
    pts = np.array(pts); pix = np.concatenate(pix) + rng.normal(0, noise_px, (n_loop * len(cams), 2))
    M0 = s["M"]
    obs_point = np.repeat(np.arange(M0 + 1, M0 + n_loop + 1, dtype=np.int64), len(cams))
    obs_pose = np.tile(cams + 1, n_loop).astype(np.int64)
    s["theta_gt"] = np.concatenate([gt, pts.ravel()])
    s["theta0"] = np.concatenate([s["theta0"], (pts + rng.normal(0, 5e-2, pts.shape)).ravel()])
    s["pixels_yx"] = np.ascontiguousarray(np.concatenate([s["pixels_yx"], pix]))
    s["pose_ids"] = np.concatenate([s["pose_ids"], obs_pose]); s["point_ids"] = np.concatenate([s["point_ids"], obs_point])
    s["M"] = M0 + n_loop; s["O"] = len(s["pose_ids"])
    return s


def ba_scene_ragged(seed):
    """A random window for fuzzing the solver's paths: 2-40 poses, 2-24 observers per point of which 0 / 15 / 40 % are dropped at
    random (ragged tracks: gaps, points with two observations), constant poses at the front, anywhere, or most of the window (the
    reference's shape), sometimes loop-closure points, sometimes a shuffled observation order."""
    rng = np.random.default_rng(seed)
    P = int(rng.integers(2, 41)); k = int(rng.integers(2, min(P, 24) + 1)); M = int(rng.integers(20, 40 * P))
    loop = P >= 16 and rng.random() < 0.3
    if loop:
        s = ba_scene_loop(P=P, M=M, seed=seed, n_loop=int(rng.integers(5, 80)), k_loop=int(rng.integers(2, 6)), obs_per_point=k)
    else:
        s = ba_scene(P=P, M=M, seed=seed, obs_per_point=k)
    M = s["M"]
    const = np.zeros(P, dtype=np.uint8)
    mode = rng.integers(0, 4)
    if mode == 0: const[0] = 1
    elif mode == 1: const[rng.random(P) < 0.3] = 1
    elif mode == 2: const[: int(rng.integers(1, P))] = 1
    else: const[rng.random(P) < 0.7] = 1
    if const.all(): const[int(rng.integers(0, P))] = 0
    # gauge: at least one constant pose unless the window is tiny
    if not const.any(): const[0] = 1
    s["theta_const"] = const
    # the perturbation of ba_scene moved only its own free poses: fine either way
    keep = rng.random(s["O"]) >= rng.choice([0.0, 0.15, 0.4])
    # every point keeps at least two observations
    cnt = np.bincount(s["point_ids"][keep], minlength=M + 1)
    short = np.isin(s["point_ids"], np.where(cnt < 2)[0])
    keep |= short
    order = np.where(keep)[0]
    if rng.random() < 0.5: order = rng.permutation(order)
    for key in ("pose_ids", "point_ids"): s[key] = s[key][order]
    s["pixels_yx"] = np.ascontiguousarray(s["pixels_yx"][order]); s["O"] = len(order)
    if P >= 6 and rng.random() < 0.4:
        # far map points seen by a random handful of poses ANYWHERE in the window (re-observed old map points, map_manager.jl:300-449):
        # the covisibility graph stops being a chain or a ring -- stars, chords -- and the pose order the solver picks is Cuthill-McKee's
        fx, fy, cx, cy = s["cam"]
        gt = s["theta_gt"][:6 * P].reshape(P, 6)
        Rp = _rot_batch(gt[:, :3]); tp = gt[:, 3:]
        n_far = int(rng.integers(1, 40)); H, W = 376, 1241
        new_pts, new_pose, new_pix = [], [], []
        for _ in range(n_far):
            a = int(rng.integers(0, P))
            u = rng.uniform(300, W - 300); v = rng.uniform(100, H - 100); z = rng.uniform(60.0, 120.0)
            Xw = Rp[a].T @ (np.array([(u - cx) / fx * z, (v - cy) / fy * z, z]) - tp[a])
            xc = np.einsum("kij,j->ki", Rp, Xw) + tp
            py = fy * xc[:, 1] / xc[:, 2] + cy; px = fx * xc[:, 0] / xc[:, 2] + cx
            vis = np.flatnonzero((xc[:, 2] > 5.0) & (py >= 1) & (py <= H) & (px >= 1) & (px <= W))
            if len(vis) < 2: continue
            cams = np.sort(rng.choice(vis, size=min(len(vis), int(rng.integers(2, 7))), replace=False))
            new_pts.append(Xw); new_pose.append(cams)
            new_pix.append(np.stack([py[cams], px[cams]], 1) + rng.normal(0, 0.5, (len(cams), 2)))
        if new_pts:
            M0 = s["M"]; k = len(new_pts)
            s["theta_gt"] = np.concatenate([s["theta_gt"], np.ravel(new_pts)])
            s["theta0"] = np.concatenate([s["theta0"], (np.array(new_pts) + rng.normal(0, 5e-2, (k, 3))).ravel()])
            s["pose_ids"] = np.concatenate([s["pose_ids"], np.concatenate(new_pose) + 1]).astype(np.int64)
            s["point_ids"] = np.concatenate([s["point_ids"], np.concatenate([np.full(len(c), M0 + j + 1) for j, c in enumerate(new_pose)])]).astype(np.int64)
            s["pixels_yx"] = np.ascontiguousarray(np.concatenate([s["pixels_yx"], np.concatenate(new_pix)]))
            s["M"] = M0 + k; s["O"] = len(s["pose_ids"])
    return s


def ba_halfband(s):
    """block half-bandwidth of the reduced camera system of a scene: widest span of FREE observers of one map point"""
    free = np.asarray(s["theta_const"])[s["pose_ids"] - 1] == 0
    pid, pose = s["point_ids"][free], s["pose_ids"][free]
    if len(pid) == 0:
        return 0
    lo = np.full(s["M"] + 1, 1 << 30); hi = np.full(s["M"] + 1, -1)
    np.minimum.at(lo, pid, pose); np.maximum.at(hi, pid, pose)
    seen = hi >= 0
    return int((hi[seen] - lo[seen]).max())


def pnp_scene(n=300, seed=0, cam=KITTI_CAM, H=376, W=1241, noise_px=0.5, outlier_frac=0.05):
    rng = np.random.default_rng(0x9A9 + seed)
    fx, fy, cx, cy = cam
    ang = (0.03, -0.05, 0.02); t = np.array([0.3, -0.1, 0.5])
    R = rotzyx(*ang)
    u = rng.uniform(20, W - 20, n); v = rng.uniform(20, H - 20, n); z = rng.uniform(5, 40, n)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = (Xc - t) @ R   # R^T (Xc - t)
    pix = np.stack([v, u], 1) + rng.normal(0, noise_px, (n, 2))
    no = int(round(outlier_frac * n)); idx = rng.choice(n, no, replace=False)
    ang_o = rng.uniform(0, 2 * np.pi, no); mag = rng.uniform(4.0, 12.0, no)
    pix[idx, 0] += mag * np.sin(ang_o); pix[idx, 1] += mag * np.cos(ang_o)
    pose_gt = np.eye(4); pose_gt[:3, :3] = R; pose_gt[:3, 3] = t
    pose0 = np.eye(4); pose0[:3, :3] = rotzyx(ang[0] + 0.01, ang[1] - 0.012, ang[2] + 0.008); pose0[:3, 3] = t + (0.05, -0.04, 0.08)
    return dict(cam=cam, pose0=pose0, pose_gt=pose_gt, pixels_yx=pix, points=Xw, gross_outliers=np.sort(idx))


def triangulation_scene(n=500, seed=0, baseline=0.54, noise_px=0.0, n_behind=0, n_gross=0, temporal=False):
    """Two pinhole views of n points in front of camera 1 (KITTI intrinsics).  stereo: camera 2 = camera 1 shifted
    by `baseline` along x (T21 = right_camera.Ti0); temporal: a general small rigid motion.  Returns a dict with
    cam (fx, fy, cx, cy), T21 (4x4), px1 / px2 (n, 2) (y, x), xyz (n, 3) ground truth in camera-1 coordinates and
    index sets `behind` (points given a pixel pair that triangulates behind the cameras) and `gross`
    (second-view pixel displaced by 30-60 px)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0x7121 + seed)
    fx, fy, cx, cy = KITTI_CAM
    X = np.stack([rng.uniform(-12, 12, n), rng.uniform(-3, 3, n), rng.uniform(4, 60, n)], axis=1)
    T = np.eye(4)
    if temporal:
        T[:3, :3] = Rotation.from_euler("ZYX", [0.01, -0.03, 0.005]).as_matrix()
        T[:3, 3] = [0.3, -0.05, -1.1]
    else:
        T[0, 3] = -baseline
    def proj(P):
        return np.stack([fy * P[:, 1] / P[:, 2] + cy, fx * P[:, 0] / P[:, 2] + cx], axis=1)
    X2 = X @ T[:3, :3].T + T[:3, 3]
    px1 = proj(X) + rng.normal(0, noise_px, (n, 2)) if noise_px else proj(X)
    px2 = proj(X2) + rng.normal(0, noise_px, (n, 2)) if noise_px else proj(X2)
    idx = rng.permutation(n)
    behind, gross = idx[:n_behind], idx[n_behind:n_behind + n_gross]
    if n_behind:                                  # swap the disparity sign: the rays meet behind the cameras
        d = px1[behind, 1] - px2[behind, 1]
        px2[behind, 1] = px1[behind, 1] + d
    if n_gross:
        px2[gross, 0] += rng.choice([-1.0, 1.0], n_gross) * rng.uniform(30, 60, n_gross)
    return dict(cam=(fx, fy, cx, cy), T21=T, px1=px1, px2=px2, xyz=X, behind=behind, gross=gross)


def p3p_scene(n=400, seed=0, noise_px=0.3, outlier_frac=0.2, iters=256):
    """Input of compute_pose! (front_end.jl:138-167): n map points seen by a KITTI camera at a known pose.
    Returns pts3d (n, 3) world, px_xy (n, 2) (x, y) as P3P expects, pdn (n, 3) normalised bearing vectors of the
    (noisy) pixels, K (3x3), Rt_gt (3x4, x_cam = R X + t), `gross` (indices given a 15-60 px displacement) and
    `samples` (iters, 3) int32 0-based distinct triples (the caller-side RNG draw)."""
    rng = np.random.default_rng(0xB3B + seed)
    fx, fy, cx, cy = KITTI_CAM
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    R = rotzyx(0.04, -0.07, 0.025); t = np.array([0.4, -0.15, 0.8])
    u = rng.uniform(20, 1221, n); v = rng.uniform(20, 356, n); z = rng.uniform(4, 50, n)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = (Xc - t) @ R
    px = np.stack([u, v], 1)
    if noise_px:
        px = px + rng.normal(0, noise_px, (n, 2))
    gross = np.sort(rng.choice(n, int(round(outlier_frac * n)), replace=False))
    ang = rng.uniform(0, 2 * np.pi, len(gross)); mag = rng.uniform(15.0, 60.0, len(gross))
    px[gross, 0] += mag * np.cos(ang); px[gross, 1] += mag * np.sin(ang)
    bear = np.stack([(px[:, 0] - cx) / fx, (px[:, 1] - cy) / fy, np.ones(n)], 1)
    pdn = bear / np.linalg.norm(bear, axis=1, keepdims=True)
    samples = np.stack([rng.permutation(n)[:3] for _ in range(iters)]).astype(np.int32)
    return dict(pts3d=np.ascontiguousarray(Xw), px_xy=np.ascontiguousarray(px), pdn=np.ascontiguousarray(pdn), K=K,
                Rt_gt=np.concatenate([R, t[:, None]], 1), gross=gross, samples=samples)


def five_point_scene(n=400, seed=0, noise_px=0.3, outlier_frac=0.2, iters=128):
    """Input of compute_pose_5pt! (front_end.jl:258-308): n keypoints seen in the previous key-frame and in the
    current frame (KITTI intrinsics, a general small rigid motion).  px1 / px2 (n, 2) pixels in (x, y) order as the
    five-point solver expects, pd1 / pd2 (n, 2) the matching normalised coordinates, K, Rt_gt (3x4, previous ->
    current, translation scaled to unit length), `gross` (second-view pixels displaced by 15-60 px) and `samples`
    (iters, 5) int32 0-based distinct 5-tuples."""
    s = triangulation_scene(n=n, seed=100 + seed, temporal=True)
    rng = np.random.default_rng(0x5F5 + seed)
    fx, fy, cx, cy = s["cam"]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    px1 = s["px1"][:, ::-1].copy(); px2 = s["px2"][:, ::-1].copy()
    if noise_px:
        px1 += rng.normal(0, noise_px, px1.shape); px2 += rng.normal(0, noise_px, px2.shape)
    gross = np.sort(rng.choice(n, int(round(outlier_frac * n)), replace=False))
    ang = rng.uniform(0, 2 * np.pi, len(gross)); mag = rng.uniform(15.0, 60.0, len(gross))
    px2[gross, 0] += mag * np.cos(ang); px2[gross, 1] += mag * np.sin(ang)
    pd1 = (px1 - [cx, cy]) / [fx, fy]; pd2 = (px2 - [cx, cy]) / [fx, fy]
    T = s["T21"]
    Rt = np.concatenate([T[:3, :3], (T[:3, 3] / np.linalg.norm(T[:3, 3]))[:, None]], 1)
    samples = np.stack([rng.permutation(n)[:5] for _ in range(iters)]).astype(np.int32)
    return dict(px1=np.ascontiguousarray(px1), px2=np.ascontiguousarray(px2), pd1=np.ascontiguousarray(pd1), pd2=np.ascontiguousarray(pd2),
                K=K, Rt_gt=Rt, gross=gross, samples=samples)
