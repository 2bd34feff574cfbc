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


def ba_scene(P=5, M=800, obs_per_point=10, seed=0, cam=KITTI_CAM, H=376, W=1241,
             noise_px=0.5, outlier_frac=0.02, n_const=1, perturb=(2e-3, 2e-2, 5e-2)):
    """Windowed BA problem in the reference's flat layout (estimator.jl:16-40):
    theta = [6P (RotZYX t1,t2,t3, t) ; 3M], pixels (O,2) as (y,x), 1-based ids.
    Cameras move forward along +z with a slight curve; each point is seen by a
    contiguous run of min(obs_per_point, P) key-frames.  Returns a dict."""
    rng = np.random.default_rng(0xBA00 + seed)
    fx, fy, cx, cy = cam
    k = min(obs_per_point, P)
    poses = np.zeros((P, 6))
    for p in range(P):
        yaw = 0.01 * p
        # world->camera: R = RotZYX(0, yaw, 0)-ish small rotation about y, camera centre advancing in z
        R = rotzyx(0.002 * p, yaw, -0.001 * p)
        c = np.array([0.05 * p * p * 0.01, 0.0, 0.8 * p])
        poses[p, :3] = (0.002 * p, yaw, -0.001 * p)
        poses[p, 3:] = -R @ c
    pts = np.zeros((M, 3))
    obs_pose, obs_point, pix = [], [], []
    for m in range(M):
        start = int(rng.integers(0, P - k + 1))
        mid = start + k // 2
        # sample a pixel + depth in the middle camera and back-project
        for _ in range(50):
            u = rng.uniform(40, W - 40); v = rng.uniform(40, H - 40); z = rng.uniform(6.0, 40.0) + 0.8 * k
            Xc = np.array([(u - cx) / fx * z, (v - cy) / fy * z, z])
            R = rotzyx(*poses[mid, :3])
            Xw = R.T @ (Xc - poses[mid, 3:])
            ok = True
            proj = []
            for p in range(start, start + k):
                Rp = rotzyx(*poses[p, :3])
                xc = Rp @ Xw + poses[p, 3:]
                if xc[2] < 1.0:
                    ok = False; break
                py = fy * xc[1] / xc[2] + cy; px_ = fx * xc[0] / xc[2] + cx
                if not (1 <= py <= H and 1 <= px_ <= W):
                    ok = False; break
                proj.append((py, px_))
            if ok:
                break
        pts[m] = Xw
        for j, p in enumerate(range(start, start + k)):
            obs_pose.append(p + 1); obs_point.append(m + 1); pix.append(proj[j] if ok else (cy, cx))
    pix = np.array(pix) + rng.normal(0, noise_px, (len(pix), 2))
    O = len(pix)
    n_out = int(round(outlier_frac * O))
    out_idx = rng.choice(O, n_out, replace=False) if n_out else np.zeros(0, dtype=int)
    # gross outliers = wrong associations a few px away (a tracker drifting onto a
    # neighbouring corner), not uniform-in-image: the reference's first LM pass is
    # not robust (bundle_adjustment.jl:41-45), so far outliers would swamp it.
    ang = rng.uniform(0, 2 * np.pi, n_out); mag = rng.uniform(4.0, 12.0, n_out)
    pix[out_idx, 0] += mag * np.sin(ang); pix[out_idx, 1] += mag * np.cos(ang)
    theta_gt = np.concatenate([poses.ravel(), pts.ravel()])
    theta0 = theta_gt.copy()
    pp = theta0[:6 * P].reshape(P, 6); lp = theta0[6 * P:].reshape(M, 3)
    const = np.zeros(P, dtype=np.uint8); const[:n_const] = 1
    free = const == 0
    pp[free, :3] += rng.normal(0, perturb[0], (free.sum(), 3))
    pp[free, 3:] += rng.normal(0, perturb[1], (free.sum(), 3))
    lp += rng.normal(0, perturb[2], lp.shape)
    # shuffle observation order so that ids are not sorted (the feeder's order is by point, then observer)
    return dict(cam=cam, P=P, M=M, O=O, theta0=theta0, theta_gt=theta_gt, theta_const=const,
                pixels_yx=np.ascontiguousarray(pix), pose_ids=np.array(obs_pose, dtype=np.int64),
                point_ids=np.array(obs_point, dtype=np.int64), gross_outliers=np.sort(out_idx))


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
