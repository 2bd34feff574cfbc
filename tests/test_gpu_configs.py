"""GPU parity on the other BASELINE.json shapes (configs[3] EuRoC-style 480x640 mono,
configs[4] synthetic FHD 1080x1920 / 4000 kpts) and the sharded-BA entry points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PLANES = ("layers", "Iy", "Ix", "Iyy", "Ixx", "Iyx")


@pytest.mark.parametrize("H,W,maxp", [(480, 640, 1000), (1080, 1920, 4000)])
def test_front_end_shapes(slam, orc, texture, H, W, maxp):
    L, R, flows = texture(H, W)
    g = []
    for im in L[:2]:
        lk = slam.LKPyramid(shape=(H, W), levels=3); slam.update_(lk, im); g.append(lk)
    ref = [orc.pyr_build(im, 3, 1.0, 1) for im in L[:2]]
    for l in (0, 3):
        for name in PLANES:
            assert np.array_equal(g[1].plane(name, l), ref[1].plane(name, l)), (name, l)
    e = slam.Extractor(maxp, 17, (-(-H // 35), -(-W // 35)), 35)
    kp = slam.detect(e, g[0], np.zeros((0, 2)))
    assert np.array_equal(kp, orc.detect(L[0], np.zeros((0, 2)), max_points=maxp))
    pts = kp[:1500].astype(float)
    out, st = slam.fb_tracking_(g[0], g[1], pts, window_size=9, pyramid_levels=3, max_distance=1.0)
    ro, rs = orc.fb_tracking(ref[0], ref[1], pts, sum_order=1, threads=4)
    assert np.array_equal(st, rs) and np.abs(out[st] - ro[st]).max() < 1e-9
    assert st.mean() > 0.7


def test_sharded_driver_on_one_gpu_matches_single_call(slam, syn):
    """world_size 1 through the multi-GPU entry points (slam_ba_create/build/solve/
    commit/flag_outliers/download) == slam_local_ba."""
    from slam_jl_amd import sharded_ba
    s = syn.ba_scene(P=12, M=1500, seed=21)
    th, ol, st = sharded_ba.sharded_bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(cache, s["cam"])
    assert np.array_equal(ol, cache.outliers)
    assert st["iters_pass1"] == cache.stats["iters_pass1"] and st["iters_pass2"] == cache.stats["iters_pass2"]
    assert abs(st["ssr_final"] - cache.stats["ssr_final"]) <= 1e-7 * cache.stats["ssr_final"], (st, cache.stats)
    assert np.abs(th - cache.theta).max() <= 1e-7 * max(1.0, np.abs(th).max())


def test_sharded_device_paced_equals_host_paced(slam, syn):
    """The device-paced pass (slam_ba_lm_* with the LM decision on the device, RCCL through slam_comm_* on the library's stream)
    against the host-paced loop over the same entry points: same iterations, outliers, cost and parameters."""
    from slam_jl_amd import sharded_ba
    s = syn.ba_scene(P=20, M=2000, seed=22)
    a = sharded_ba.sharded_bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    b = sharded_ba.sharded_bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], host_paced=True)
    assert np.array_equal(a[1], b[1])
    assert a[2]["iters_pass1"] == b[2]["iters_pass1"] and a[2]["iters_pass2"] == b[2]["iters_pass2"]
    for k in ("ssr_init", "ssr_pass1", "ssr_final"):
        assert abs(a[2][k] - b[2][k]) <= 1e-9 * b[2][k], k
    assert np.abs(a[0] - b[0]).max() <= 1e-9 * max(1.0, np.abs(b[0]).max())


def test_comm_collectives_single_rank(slam):
    """slam_comm_* (RCCL bound at run time) with one rank: all-reduce and all-gather are identities."""
    import ctypes as C
    import torch
    ctx = slam.default_context(0)
    idbuf = C.create_string_buffer(128)
    ctx.check(ctx.lib.slam_comm_unique_id(idbuf))
    h = C.c_void_p()
    ctx.check(ctx.lib.slam_comm_create(ctx.h, 1, 0, idbuf, C.byref(h)))
    assert ctx.lib.slam_comm_size(h) == 1 and ctx.lib.slam_comm_rank(h) == 0
    a = torch.arange(1000, dtype=torch.float64, device="cuda"); b = torch.zeros(4, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.check(ctx.lib.slam_comm_allreduce_sum(ctx.h, h, C.c_void_p(a.data_ptr()), 1000))
    ctx.check(ctx.lib.slam_comm_allgather(ctx.h, h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 4))
    ctx.synchronize()
    assert torch.equal(a.cpu(), torch.arange(1000, dtype=torch.float64)) and torch.equal(b.cpu(), torch.arange(4, dtype=torch.float64))
    ctx.check(ctx.lib.slam_comm_destroy(h))


def test_profiling_spans(slam, texture):
    ctx = slam.default_context(0)
    L = texture(120, 160)[0]
    lk = slam.LKPyramid(shape=(120, 160), levels=3)
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(3):
        slam.update_(lk, L[0])
    ms, n = ctx.prof_get("pyr_update"); ms2, n2 = ctx.prof_get("k_iir_rows")
    ctx.prof_enable(False)
    assert n == 3 and n2 == 12 and 0 < ms2 < ms
    slam.update_(lk, L[1])                                     # graph path gives the same planes as the span path
    a = lk.plane("Iyx", 2)
    ctx.prof_enable(True); slam.update_(lk, L[1]); ctx.prof_enable(False)
    assert np.array_equal(a, lk.plane("Iyx", 2))
