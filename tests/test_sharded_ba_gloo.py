"""CPU, world_size 2 over gloo: the point-sharded LM driver (partitioning, the
per-iteration all-reduce of [S; g; diag; ssr], identical accept/reject on all
ranks, result gathering) against the single-process oracle.  The compute shard
is tests/np_ba.NumpyShard -- the product's HipShard needs a GPU; its kernels are
covered by the -m gpu tests through the same slam_ba_* entry points."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import slam_jl_amd  # noqa: F401
    from slam_jl_amd import sharded_ba, synthetic as syn
    import np_ba
    s = syn.ba_scene(P=6, M=90, seed=11, obs_per_point=4)
    th, ol, st = sharded_ba.sharded_bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"],
                                                      s["point_ids"], shard_factory=np_ba.NumpyShard)
    q.put((rank, th, ol, st))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_points_balanced():
    import slam_jl_amd  # noqa: F401
    from slam_jl_amd import sharded_ba
    pid = np.repeat(np.arange(1, 101), np.random.default_rng(0).integers(1, 9, 100))
    for ws in (1, 2, 3, 8):
        parts = sharded_ba.partition_points(pid, 100, ws)
        assert parts[0][0] == 0 and parts[-1][1] == 100 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        cnt = [np.isin(pid - 1, np.arange(lo, hi)).sum() for lo, hi in parts]
        assert max(cnt) - min(cnt) <= 16
    assert not sharded_ba.worth_sharding(5, 4000, 8) and not sharded_ba.worth_sharding(50, 100000, 1)


def test_single_process_driver_matches_oracle(orc, syn):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import np_ba
    from slam_jl_amd import sharded_ba
    s = syn.ba_scene(P=6, M=90, seed=11, obs_per_point=4)
    th, ol, st = sharded_ba.sharded_bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"],
                                                      s["point_ids"], shard_factory=np_ba.NumpyShard)
    rt, ro, rs = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
    assert np.array_equal(ol, ro) and st["iters_pass1"] == rs["iters_pass1"] and st["iters_pass2"] == rs["iters_pass2"]
    assert abs(st["ssr_final"] - rs["ssr_final"]) < 1e-7 * rs["ssr_final"]
    assert np.abs(th - rt).max() < 1e-6


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_oracle(orc, syn):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s = syn.ba_scene(P=6, M=90, seed=11, obs_per_point=4)
    rt, ro, rs = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
    (_, t0, o0, s0), (_, t1, o1, s1) = res
    assert np.array_equal(t0, t1) and np.array_equal(o0, o1)                  # every rank returns the same full result
    assert s0["world_size"] == 2 and s0["points_local"] + s1["points_local"] == 90
    assert np.array_equal(o0, ro)
    assert abs(s0["ssr_final"] - rs["ssr_final"]) < 1e-7 * rs["ssr_final"]
    assert np.abs(t0 - rt).max() < 1e-6


@pytest.mark.timeout(400)
def test_four_rank_gloo_matches_oracle(orc, syn):
    """the same with four ranks (the driver's scaling runs go to 8): four point ranges, the all-reduce over four contributions, every rank the same result"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(4)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s = syn.ba_scene(P=6, M=90, seed=11, obs_per_point=4)
    rt, ro, rs = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])
    assert res[0][3]["world_size"] == 4 and sum(r[3]["points_local"] for r in res) == 90
    assert np.array_equal(res[0][2], ro)
    assert abs(res[0][3]["ssr_final"] - rs["ssr_final"]) < 1e-7 * rs["ssr_final"]
    assert np.abs(res[0][1] - rt).max() < 1e-6
