"""GPU: the device-paced point-sharded LM pass (slam_ba_lm_* with nranks = 2) on ONE GPU in ONE process.

RCCL refuses two ranks on one device and no multi-GPU box is available to the build, so the two collectives of the protocol are
emulated where they sit in the call sequence: the all-reduce of the reduce buffers is a device add of the two shards' buffers,
the all-gather of the trial costs a concatenation in rank order.  Everything else is the product path of
slam.jl_amd/sharded_ba.py::HipShard.lm_pass -- slam_ba_lm_begin / _start / _build / _solve / _step(nranks = 2) / _state -- so the
rank-order fold of k_control_gathered, the on-device accept / reject and the commit run with two ranks' worth of inputs.
Both shards must take identical decisions and the result must equal slam_local_ba on the whole problem."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _two_shard_ba(slam, s, bounds, iters_fast=5, iterations=10, repr_eps=5.0):
    import torch
    from slam_jl_amd import sharded_ba
    from slam_jl_amd import _lib as L
    ctx = slam.default_context(0)
    lib = ctx.lib
    P = s["P"]; n = 6 * P
    theta, tc, px, pi, li = s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"]
    shards, sels = [], []
    for (lo, hi) in bounds:
        sel = np.where((li - 1 >= lo) & (li - 1 < hi))[0]
        th = np.concatenate([theta[:n], theta[n + 3 * lo:n + 3 * hi]])
        shards.append(sharded_ba.HipShard(s["cam"], P, th, tc, px[sel], pi[sel], li[sel] - lo, ctx=ctx)); sels.append(sel)
    hbs = [sh.halfband() for sh in shards]
    for sh in shards:
        sh.set_halfband(max(hbs))
    gathered = torch.zeros(8, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    red = [C.c_void_p(sh.red.data_ptr()) for sh in shards]
    trial = [C.c_void_p(sh.trial.data_ptr()) for sh in shards]

    def all_reduce():
        ctx.synchronize()
        tot = shards[0].red + shards[1].red
        shards[0].red.copy_(tot); shards[1].red.copy_(tot)
        torch.cuda.synchronize()

    def all_gather():
        ctx.synchronize()
        gathered.copy_(torch.cat([shards[0].trial, shards[1].trial]))
        torch.cuda.synchronize()

    def state(k):
        st = np.zeros(8)
        ctx.check(lib.slam_ba_lm_state(ctx.h, shards[k].h, L.ptr(st)))
        return st

    def lm_pass(ignore, iters, first):
        for k in range(2):
            ctx.check(lib.slam_ba_lm_begin(ctx.h, shards[k].h, ignore, red[k]))
        all_reduce()
        for k in range(2):
            ctx.check(lib.slam_ba_lm_start(ctx.h, shards[k].h, red[k], first))
        for it in range(1, iters + 1):
            if it > 1:
                for k in range(2):
                    ctx.check(lib.slam_ba_lm_build(ctx.h, shards[k].h, ignore, red[k]))
                all_reduce()
            for k in range(2):
                ctx.check(lib.slam_ba_lm_solve(ctx.h, shards[k].h, red[k], ignore, trial[k]))
            all_gather()
            for k in range(2):
                ctx.check(lib.slam_ba_lm_step(ctx.h, shards[k].h, C.c_void_p(gathered.data_ptr()), 2, it))
            a, b = state(0), state(1)
            assert np.array_equal(a, b), ("the two shards disagree after iteration", it, a, b)
        return state(0)

    st1 = lm_pass(0, iters_fast, 1)
    assert st1[4] == 0.0
    n_out = sum(sh.flag_outliers(repr_eps) for sh in shards)
    st2 = lm_pass(1, iterations, 0)
    assert st2[4] == 0.0
    theta_out = theta.copy(); outl = np.zeros(len(pi), bool)
    for k, (lo, hi) in enumerate(bounds):
        th, ol = shards[k].download()
        if k == 0:
            theta_out[:n] = th[:n]
        else:
            assert np.array_equal(theta_out[:n], th[:n]), "the shards' poses differ"
        theta_out[n + 3 * lo:n + 3 * hi] = th[n:]
        outl[sels[k]] = ol
    for sh in shards:
        sh.close()
    return theta_out, outl, dict(ssr_init=st1[5], ssr_final=st2[0], iters_pass1=int(st1[1]), iters_pass2=int(st2[1]), n_outliers=n_out, hbs=hbs)


def _single(slam, s):
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(cache, s["cam"])
    return cache


@pytest.mark.parametrize("P,M", [(20, 2000), (50, 10000)])
def test_two_shards_on_one_gpu_equal_the_single_gpu_solve(slam, syn, P, M):
    from slam_jl_amd import sharded_ba
    s = syn.ba_scene(P=P, M=M, seed=21)
    bounds = sharded_ba.partition_points(s["point_ids"], s["M"], 2)
    theta, outl, st = _two_shard_ba(slam, s, bounds)
    ref = _single(slam, s)
    assert st["iters_pass1"] == ref.stats["iters_pass1"] and st["iters_pass2"] == ref.stats["iters_pass2"]
    assert np.array_equal(outl, ref.outliers) and st["n_outliers"] == ref.stats["n_outliers"]
    assert abs(st["ssr_final"] - ref.stats["ssr_final"]) <= 1e-8 * ref.stats["ssr_final"]
    assert np.abs(theta - ref.theta).max() <= 1e-6


def test_two_shards_with_different_half_bandwidths(slam, syn):
    """shard 0's points have 10 consecutive observers (half-bandwidth 9), shard 1's only 4 (half-bandwidth 3).  The
    all-reduced system has the wider band; the narrow shard's reduce buffer must not keep the previous iteration's sum in the blocks
    outside its own band (it is zeroed on every build when it is caller-owned), and it must solve with the agreed maximum."""
    P = 24
    a = syn.ba_scene(P=P, M=1500, seed=31, obs_per_point=10)
    b = syn.ba_scene(P=P, M=1500, seed=32, obs_per_point=4)
    n = 6 * P
    assert np.array_equal(a["theta_gt"][:n], b["theta_gt"][:n])            # same track
    s = dict(cam=a["cam"], P=P, M=a["M"] + b["M"], theta_const=a["theta_const"],
             theta0=np.concatenate([a["theta0"], b["theta0"][n:]]),
             pixels_yx=np.concatenate([a["pixels_yx"], b["pixels_yx"]]),
             pose_ids=np.concatenate([a["pose_ids"], b["pose_ids"]]),
             point_ids=np.concatenate([a["point_ids"], b["point_ids"] + a["M"]]))
    s["O"] = len(s["pose_ids"])
    bounds = [(0, a["M"]), (a["M"], s["M"])]
    theta, outl, st = _two_shard_ba(slam, s, bounds)
    assert st["hbs"] == [9, 3]
    ref = _single(slam, s)
    assert st["iters_pass1"] == ref.stats["iters_pass1"] and st["iters_pass2"] == ref.stats["iters_pass2"]
    assert np.array_equal(outl, ref.outliers)
    assert abs(st["ssr_final"] - ref.stats["ssr_final"]) <= 1e-8 * ref.stats["ssr_final"]
    assert np.abs(theta - ref.theta).max() <= 1e-6
    # the same split through the HOST-paced loop of sharded_ba (slam_ba_build / _solve / _commit per shard, numpy decisions):
    # ADVICE r2 (medium) -- it used to solve with each shard's local band


def test_two_shards_on_ragged_windows(slam, syn):
    """Ragged windows (syn.ba_scene_ragged: dropped observations, constant poses anywhere, loop closures) cut into two shards at an
    arbitrary point: the shards' local bands and solver paths differ from each other's and from the single solve's (which may reorder
    the poses), early convergence leaves iterations that do nothing -- decisions, outliers, cost and parameters must still agree.
    (Seeds 76 ... 145 converge early: the converged state used to pick up each shard's stale LOCAL trial cost.)
    tests/fuzz/ba_shard_fuzz.py runs more."""
    from slam_jl_amd import sharded_ba
    for seed in (3, 59, 76, 98, 110, 125, 133, 138, 145):
        s = syn.ba_scene_ragged(seed)
        rng = np.random.default_rng(seed)
        cut = int(rng.integers(1, s["M"] - 1)) if rng.random() < 0.5 else None
        bounds = [(0, cut), (cut, s["M"])] if cut else sharded_ba.partition_points(s["point_ids"], s["M"], 2)
        theta, outl, st = _two_shard_ba(slam, s, bounds)
        ref = _single(slam, s)
        assert (st["iters_pass1"], st["iters_pass2"]) == (ref.stats["iters_pass1"], ref.stats["iters_pass2"]), seed
        assert np.array_equal(outl, ref.outliers), seed
        assert abs(st["ssr_final"] - ref.stats["ssr_final"]) <= 1e-8 * ref.stats["ssr_final"], seed
        assert np.abs(theta - ref.theta).max() <= 1e-6 * max(1.0, np.abs(ref.theta).max()), seed


def test_one_rank_collectives_are_timed(slam, syn):
    """ba_sharded's `collectives_us` (the measured half of worth_sharding's constants): one-rank RCCL all-reduce of the reduce
    buffer + all-gather of the trial costs on the library stream."""
    from slam_jl_amd import sharded_ba
    s = syn.ba_scene(P=20, M=1000, seed=3)
    _, _, st = sharded_ba.sharded_bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], timings={})
    c = st["collectives_us"]
    assert c is not None and 0 < c["allreduce"] < 5e3 and 0 < c["allgather"] < 5e3 and c["allreduce_bytes"] == (120 * 120 + 240 + 8) * 8
