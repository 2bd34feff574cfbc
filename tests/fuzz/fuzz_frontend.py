"""Random shapes / parameters through the front-end kernels against the oracle (what the fixed-shape tests do not visit):
  pyr     single-image pyramids, random H x W (4 .. 300), levels, sigma, both ctor modes
  batch   batched pyramids with the bandwidth-bound kernel set FORCED on small shapes (SLAMHIP_CK_MIN_MB=0: k_cols_fused, k_iir_rows_ck
          with the fused resize, k_cum_fused), f64 / u8 ingest, target-only builds, S = 1 .. 6
  tolpyr  single-image tolerance mode (mode 3) on random shapes up to 400 x 1400: planes <= 1e-11 relative to the oracle's exact build
  tolbatch the tolerance-mode batch build (mode 3, S = 4 .. 10) forced onto small random shapes: planes <= 1e-11 relative to the oracle's exact build
  lk      fb_tracking with random window sizes (2 .. 14: the three cached instantiations and the uncached path), levels, priors, points on and near the borders
  detect  random shapes, cell sizes, current keypoints (none / few / many / clustered), mask sigma
  brief   describe with random shapes and keypoints on / next to the borders (dropped ones included)
python tests/fuzz/fuzz_frontend.py [n per part] [seed0] [parts]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
from oracle import oracle as orc
PLANES = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
parts = sys.argv[3].split(",") if len(sys.argv) > 3 else ["pyr", "batch", "tolpyr", "tolbatch", "lk", "detect", "brief"]
fails = 0


def fail(msg):
    global fails
    fails += 1
    print("FAIL", msg, flush=True)


def rand_image(rng, H, W):
    kind = rng.integers(0, 3)
    if kind == 0: return np.asfortranarray(rng.random((H, W)))
    if kind == 1: return np.asfortranarray(np.round(rng.random((H, W)) * 255) / 255.0)          # 8-bit values
    base = syn.texture_canvas(H, W, seed=int(rng.integers(0, 1000)), margin=0) if min(H, W) >= 16 else rng.random((H, W))
    return np.asfortranarray(np.clip(base + 0.01 * rng.standard_normal((H, W)), 0, 1))


def max_levels(H, W):
    l = 0
    while l < 3 and min(-(-H // 2 ** (l + 1)), -(-W // 2 ** (l + 1))) >= 4: l += 1
    return l


if "pyr" in parts:
    for t in range(n):
        rng = np.random.default_rng(seed0 + t)
        H, W = int(rng.integers(4, 300)), int(rng.integers(4, 300))
        levels = int(rng.integers(0, max_levels(H, W) + 1)); sigma = float(rng.choice([0.6, 1.0, 1.7, 2.5])); mode = int(rng.integers(0, 2))
        img = rand_image(rng, H, W)
        try:
            lk = slam.LKPyramid(img, levels, sigma=sigma) if mode == 0 else slam.LKPyramid(shape=(H, W), levels=levels)
            if mode == 1: slam.update_(lk, img, sigma=sigma)
            ref = orc.pyr_build(img, levels, sigma, mode)
            for l in range(levels + 1):
                if lk.level_shape(l) != (ref.Hs[l], ref.Ws[l]): fail(f"pyr seed {seed0 + t} {H}x{W} level {l} shape {lk.level_shape(l)} vs {(ref.Hs[l], ref.Ws[l])}")
                for name in PLANES:
                    if not np.array_equal(lk.plane(name, l), ref.plane(name, l)): fail(f"pyr seed {seed0 + t} {H}x{W} levels {levels} sigma {sigma} mode {mode}: {name} level {l}"); break
            lk.close()
        except Exception as ex:
            fail(f"pyr seed {seed0 + t} {H}x{W} levels {levels}: {repr(ex)[:200]}")
    print("pyr done", flush=True)

if "batch" in parts:
    for t in range(n):
        rng = np.random.default_rng(10000 + seed0 + t)
        H, W = int(rng.integers(8, 220)), int(rng.integers(8, 300))
        levels = int(rng.integers(0, max_levels(H, W) + 1)); S = int(rng.integers(1, 7)); u8 = bool(rng.integers(0, 2)); tgt = bool(rng.integers(0, 2)); forced = bool(rng.integers(0, 4))
        imgs = [rand_image(rng, H, W) for _ in range(S)]
        if u8:
            raw = [np.round(im * 255).astype(np.uint8) for im in imgs]
            imgs = [np.asfortranarray(r.astype(np.float64) / 255.0) for r in raw]
            dev = [torch.from_numpy(np.ascontiguousarray(r.T)).cuda() for r in raw]
        else:
            dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
        torch.cuda.synchronize()
        tag = f"batch seed {10000 + seed0 + t} {H}x{W} levels {levels} S {S} u8 {u8} target_only {tgt} forced {forced}"
        try:
            if forced: os.environ["SLAMHIP_CK_MIN_MB"] = "0"
            b = slam.PyramidBatch((H, W), levels=levels, S=S)
            b.update_([d.data_ptr() for d in dev], u8=u8, target_only=tgt)
            b.update_([d.data_ptr() for d in dev], u8=u8, target_only=tgt)          # graph replay
            os.environ.pop("SLAMHIP_CK_MIN_MB", None)
            for s_ in {0, S - 1}:
                ref = orc.pyr_build(imgs[s_], levels, 1.0, 1)
                for l in range(levels + 1):
                    for name in (PLANES if (l == 0 or not tgt) else ("layers",)):
                        if not np.array_equal(b.pyramids[s_].plane(name, l), ref.plane(name, l)): fail(f"{tag}: stream {s_} {name} level {l}"); break
        except Exception as ex:
            os.environ.pop("SLAMHIP_CK_MIN_MB", None)
            fail(f"{tag}: {repr(ex)[:200]}")
    print("batch done", flush=True)

if "tolpyr" in parts:
    # single-image tolerance mode (slam_pyr_update mode 3: k_iir_seg / k_cum_seg along y, k_rows_tol along x where the row is long enough)
    for t in range(n):
        rng = np.random.default_rng(60000 + seed0 + t)
        H, W = int(rng.integers(4, 400)), int(rng.integers(4, 1400))
        levels = int(rng.integers(0, max_levels(H, W) + 1))
        img = rand_image(rng, H, W)
        tag = f"tolpyr seed {60000 + seed0 + t} {H}x{W} levels {levels}"
        try:
            lk = slam.LKPyramid(shape=(H, W), levels=levels)
            slam.update_(lk, img, fast=True)
            slam.update_(lk, img, fast=True)                       # graph replay
            ref = orc.pyr_build(img, levels, 1.0, 1)
            for l in range(levels + 1):
                for name in PLANES:
                    g, r = lk.plane(name, l), ref.plane(name, l)
                    err = np.abs(g - r).max() / max(np.abs(r).max(), 1e-300)
                    if not (err <= 1e-11): fail(f"{tag}: {name} level {l} rel {err:.2e}"); break
        except Exception as ex:
            fail(f"{tag}: {repr(ex)[:200]}")
    print("tolpyr done", flush=True)

if "tolbatch" in parts:
    # the tolerance-mode batch build (mode 3, S >= 4: k_cols_fused<TOL[, DEC]> + k_rows_tol) forced onto small random shapes: every plane within
    # 1e-11 of the oracle's exact build relative to the plane's magnitude (heights >= 64 reach the fused kernels; odd / even heights and pitches
    # select the halved or the full blur plane; widths select the samples per thread of the row kernel)
    for t in range(n):
        rng = np.random.default_rng(50000 + seed0 + t)
        H, W = int(rng.integers(64, 330)), int(rng.integers(12, 520))
        levels = int(rng.integers(0, max_levels(H, W) + 1)); S = int(rng.integers(4, 11)); u8 = bool(rng.integers(0, 2)); tgt = bool(rng.integers(0, 3) == 0)
        imgs = [rand_image(rng, H, W) for _ in range(S)]
        if u8:
            raw = [np.round(im * 255).astype(np.uint8) for im in imgs]
            imgs = [np.asfortranarray(r.astype(np.float64) / 255.0) for r in raw]
            dev = [torch.from_numpy(np.ascontiguousarray(r.T)).cuda() for r in raw]
        else:
            dev = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in imgs]
        torch.cuda.synchronize()
        tag = f"tolbatch seed {50000 + seed0 + t} {H}x{W} levels {levels} S {S} u8 {u8} target_only {tgt}"
        try:
            os.environ["SLAMHIP_CK_MIN_MB"] = "0"
            b = slam.PyramidBatch((H, W), levels=levels, S=S)
            b.update_([d.data_ptr() for d in dev], u8=u8, target_only=tgt, fast=True)
            b.update_([d.data_ptr() for d in dev], u8=u8, target_only=tgt, fast=True)          # graph replay
            os.environ.pop("SLAMHIP_CK_MIN_MB", None)
            for s_ in {0, S - 1}:
                ref = orc.pyr_build(imgs[s_], levels, 1.0, 1)
                for l in range(levels + 1):
                    for name in (PLANES if (l == 0 or not tgt) else ("layers",)):
                        g, r = b.pyramids[s_].plane(name, l), ref.plane(name, l)
                        err = np.abs(g - r).max() / max(np.abs(r).max(), 1e-300)
                        if not (err <= 1e-11): fail(f"{tag}: stream {s_} {name} level {l} rel {err:.2e}"); break
        except Exception as ex:
            os.environ.pop("SLAMHIP_CK_MIN_MB", None)
            fail(f"{tag}: {repr(ex)[:200]}")
    print("tolbatch done", flush=True)

if "lk" in parts:
    for t in range(n):
        rng = np.random.default_rng(20000 + seed0 + t)
        H, W = int(rng.integers(40, 260)), int(rng.integers(40, 320))
        levels = int(rng.integers(0, max_levels(H, W) + 1)); window = int(rng.integers(2, 15)); maxd = float(rng.choice([0.5, 1.0, 3.0]))
        step = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)))
        L, R, flows = syn.stereo_stream((H, W), 2, int(rng.integers(0, 50)), step, 5.0)
        npts = int(rng.integers(1, 400))
        pts = np.stack([rng.uniform(1, H, npts), rng.uniform(1, W, npts)], 1)
        edge = rng.random(npts) < 0.2                                        # on / next to the borders
        pts[edge, 0] = rng.choice([1.0, 1.5, 2.0, H - 1.0, H - 0.5, float(H)], edge.sum())
        edge = rng.random(npts) < 0.2
        pts[edge, 1] = rng.choice([1.0, 1.5, 2.0, W - 1.0, W - 0.5, float(W)], edge.sum())
        disp = None if rng.integers(0, 2) else np.tile(np.array(flows[1]) / 2.0, (npts, 1)) + rng.normal(0, 0.3, (npts, 2))
        tag = f"lk seed {20000 + seed0 + t} {H}x{W} levels {levels} window {window} points {npts} prior {disp is not None}"
        try:
            g = []
            for im in L[:2]:
                lk = slam.LKPyramid(shape=im.shape, levels=levels); slam.update_(lk, im); g.append(lk)
            r = [orc.pyr_build(im, levels, 1.0, 1) for im in L[:2]]
            out, st = slam.fb_tracking_(g[0], g[1], pts, displacement=disp, pyramid_levels=levels, window_size=window, max_distance=maxd)
            o1, s1 = orc.fb_tracking(r[0], r[1], pts, disp, 30, window, levels, 1e-4, 1e-2, maxd, sum_order=1)
            if not np.array_equal(st, s1): fail(f"{tag}: status differs at {np.where(st != s1)[0][:5]} of {npts}")
            elif st.any() and np.abs(out[st] - o1[st]).max() > 1e-9: fail(f"{tag}: positions {np.abs(out[st] - o1[st]).max():.2e} px")
            for lk in g: lk.close()
        except Exception as ex:
            fail(f"{tag}: {repr(ex)[:200]}")
    print("lk done", flush=True)

if "detect" in parts:
    for t in range(n):
        rng = np.random.default_rng(30000 + seed0 + t)
        cell = int(rng.integers(12, 52)); H, W = int(rng.integers(cell + 2, 400)), int(rng.integers(cell + 2, 500))
        maxp = int(rng.integers(1, 1500)); sig = float(rng.choice([0.0, 1.0, 3.0]))
        img = rand_image(rng, H, W)
        kind = rng.integers(0, 4)
        ncur = [0, int(rng.integers(1, 20)), int(rng.integers(20, 1200)), int(rng.integers(20, 300))][kind]
        cur = np.stack([rng.uniform(0.5, H + 0.49, ncur), rng.uniform(0.5, W + 0.49, ncur)], 1) if ncur else np.zeros((0, 2))
        if kind == 3: cur = np.clip(np.array([H / 2, W / 2]) + rng.normal(0, 6, (ncur, 2)), 1, [H, W])      # clustered
        e = slam.Extractor(maxp, max(5, cell // 2), (-(-H // cell), -(-W // cell)), cell)
        tag = f"detect seed {30000 + seed0 + t} {H}x{W} cell {cell} max_points {maxp} current {ncur} sigma {sig}"
        try:
            got = slam.detect(e, img, cur, sigma_mask=sig)
            ref = orc.detect(img, cur, max_points=maxp, radius=e.radius, cell_size=cell, sigma_mask=sig)
            if not np.array_equal(got, ref): fail(f"{tag}: {len(got)} vs {len(ref)} keypoints")
        except Exception as ex:
            fail(f"{tag}: {repr(ex)[:200]}")
    print("detect done", flush=True)
if "brief" in parts:
    pat = slam.brief_pattern()
    for t in range(n):
        rng = np.random.default_rng(40000 + seed0 + t)
        H, W = int(rng.integers(30, 300)), int(rng.integers(30, 400))
        img = rand_image(rng, H, W)
        nk = int(rng.integers(0, 400))
        kp = np.stack([rng.integers(1, H + 1, nk), rng.integers(1, W + 1, nk)], 1).astype(np.int64)
        edge = rng.random(nk) < 0.3
        kp[edge, 0] = rng.choice([1, 2, 15, 16, 17, H - 17, H - 16, H - 15, H - 1, H], edge.sum())
        edge = rng.random(nk) < 0.3
        kp[edge, 1] = rng.choice([1, 2, 15, 16, 17, W - 17, W - 16, W - 15, W - 1, W], edge.sum())
        tag = f"brief seed {40000 + seed0 + t} {H}x{W} keypoints {nk}"
        try:
            bits, rc = slam.describe(slam.Extractor(1000, 17, (-(-H // 35), -(-W // 35)), 35), img, kp, pattern=pat)
            rbits, rrc = orc.describe(img, kp, pat)
            if not (np.array_equal(rc, rrc) and np.array_equal(bits, rbits)): fail(f"{tag}: {len(rc)} vs {len(rrc)} described")
        except Exception as ex:
            fail(f"{tag}: {repr(ex)[:200]}")
    print("brief done", flush=True)
print(f"{n} trials per part, parts {parts}: {fails} failures")
