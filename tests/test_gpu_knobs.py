"""GPU: the build variants behind environment knobs stay bit-exact / within tolerance (each knob is read once per process, so every case
runs in a process of its own), and contexts recycle their streams."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECK = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
from oracle import oracle as orc
PL = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")
H, W, S = %(H)d, %(W)d, %(S)d
rng = np.random.default_rng(5)
base = syn.texture_canvas(H, W, seed=3, margin=0)
fr = [np.asfortranarray(np.round(np.clip(base + 0.03 * rng.standard_normal((H, W)), 0, 1) * 255).astype(np.uint8)) for _ in range(S)]
if S > 1:
    dev = torch.from_numpy(np.stack([np.ascontiguousarray(f.T) for f in fr])).cuda(); torch.cuda.synchronize()
    pb = slam.PyramidBatch((H, W), levels=3, S=S)
    for fast in (False, True):
        pb.update_([dev.data_ptr() + s * H * W for s in range(S)], u8=True, fast=fast); pb.update_([dev.data_ptr() + s * H * W for s in range(S)], u8=True, fast=fast)
        for s in (0, S - 1):
            ref = orc.pyr_build(np.asfortranarray(fr[s].astype(np.float64) / 255.0), 3, 1.0, 1)
            for l in range(4):
                for nm in PL:
                    g, r = pb.pyramids[s].plane(nm, l), ref.plane(nm, l)
                    ok = np.abs(g - r).max() <= 1e-11 * max(np.abs(r).max(), 1e-300) if fast else np.array_equal(g, r)
                    assert ok, (fast, s, nm, l)
else:
    img = np.asfortranarray(fr[0].astype(np.float64) / 255.0)
    for mode in (0, 1):
        p = slam.LKPyramid(img, 3) if mode == 0 else slam.LKPyramid(shape=(H, W), levels=3)
        if mode == 1:
            slam.update_(p, img); slam.update_(p, img)
        ref = orc.pyr_build(img, 3, 1.0, mode)
        for l in range(4):
            for nm in PL:
                assert np.array_equal(p.plane(nm, l), ref.plane(nm, l)), (mode, nm, l)
    p = slam.LKPyramid(shape=(H, W), levels=3)                 # tolerance mode of a single image (segmented recurrences)
    slam.update_(p, img, fast=True); slam.update_(p, img, fast=True)
    ref = orc.pyr_build(img, 3, 1.0, 1)
    for l in range(4):
        for nm in PL:
            g, r = p.plane(nm, l), ref.plane(nm, l)
            assert np.abs(g - r).max() <= 1e-11 * max(np.abs(r).max(), 1e-300), ("mode 3", nm, l, np.abs(g - r).max())
print("OK")
'''


def _run(env, H, W, S):
    code = CHECK % dict(root=ROOT, H=H, W=W, S=S)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-800:] + r.stderr[-1500:]


def test_sub_batched_build_is_bit_exact_and_within_tolerance():
    """SLAMHIP_PYR_CHUNK_MB: the batch built in sub-batches of the big levels (measured slower, kept as a knob) gives the same planes"""
    _run({"SLAMHIP_PYR_CHUNK_MB": "8", "SLAMHIP_CK_MIN_MB": "1"}, 200, 320, 24)


def test_latency_topology_of_a_single_image_is_bit_exact():
    """SLAMHIP_TOPOLOGY=1: the blur chain alone on the main lane, one graph branch per level -- same planes, both border modes"""
    _run({"SLAMHIP_TOPOLOGY": "1"}, 370, 1226, 1)


def test_single_image_scharr_columns_per_thread_are_bit_exact():
    for xc in ("1", "4", "16"):
        _run({"SLAMHIP_SCHARR_XC1": xc}, 121, 163, 1)


def test_single_image_segment_variants_stay_within_tolerance():
    """mode 3, one image: the 128-segment column kernels (SLAMHIP_SEG_WIDE), the running sum along y as a launch of its own
    (SLAMHIP_NO_SEG_CUM) and the defaults (32 segments of <= 16 samples, running sum fused), at sizes on both sides of the
    32 x 16-sample limit"""
    for env in ({}, {"SLAMHIP_SEG_WIDE": "1"}, {"SLAMHIP_NO_SEG_CUM": "1"}, {"SLAMHIP_SEG_WIDE": "1", "SLAMHIP_NO_SEG_CUM": "1"}):
        _run(env, 370, 1226, 1)
    _run({}, 540, 700, 1)                                   # columns of 540 samples: 32 segments would need 17 samples -> the 128-segment variant
    _run({}, 121, 163, 1)


def test_contexts_recycle_their_streams(slam):
    """slam_ctx_destroy parks the stream; a later context of the same scheduling class takes it over (events other libraries recorded on
    it stay valid), a context of another class never does"""
    lo = slam.Context(0, priority=-1); slo = lo.stream; lo.close()
    a = slam.Context(0); sa = a.stream; a.close()
    made, seen = [], set()
    for _ in range(256):                                   # (earlier tests may have parked streams of their own: they come out first)
        c = slam.Context(0); made.append(c); seen.add(c.stream)
        if c.stream == sa:
            break
    assert sa in seen and slo not in seen
    lo2 = slam.Context(0, priority=-1)
    got_lo = {lo2.stream}
    extra = []
    while slo not in got_lo and len(extra) < 64:
        e = slam.Context(0, priority=-1); extra.append(e); got_lo.add(e.stream)
    assert slo in got_lo
    for x in made + extra + [lo2]:
        x.close()
