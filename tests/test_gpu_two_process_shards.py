"""GPU: TWO REAL PROCESSES through the product's HipShard on one GPU (north_star: "the BA normal-equation build shards
observations across GPUs").  RCCL refuses two ranks on one device, so the two ranks share GPU 0 and run the HOST-paced path of
sharded_bundle_adjustment (slam_ba_build / slam_ba_solve / slam_ba_commit per rank, the two collectives of an LM iteration over
torch.distributed's gloo backend): the partition by map point, the per-iteration all-reduce of [S; g; diag; ssr], the agreed
half-bandwidth, identical accept / reject decisions on both ranks and the gathered result are the product code; only the transport
differs from a multi-GPU node.  The result must equal slam_local_ba on the whole window (outlier sets equal, cost 1e-8, theta 1e-6).
A second case kills one rank: the survivor must leave its collective and exit non-zero within the deadline.
(The device-paced RCCL path -- slam_ba_lm_* + slam_comm_* -- has only ever run with ONE rank: no multi-GPU hardware was available
to the build; no scaling curve exists.)"""
import os
import socket
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene(syn, name):
    if name == "P20":
        return syn.ba_scene(P=20, M=2000, seed=21)
    if name == "P50":
        return syn.ba_scene(P=50, M=10000, seed=21)
    # "unequal": rank 0's points have 10 consecutive observers (half-bandwidth 9), rank 1's only 4 (half-bandwidth 3); equal observation counts
    P = 24
    a = syn.ba_scene(P=P, M=600, seed=31, obs_per_point=10)
    b = syn.ba_scene(P=P, M=1500, seed=32, obs_per_point=4)
    n = 6 * P
    s = dict(cam=a["cam"], P=P, M=a["M"] + b["M"], theta_const=a["theta_const"],
             theta0=np.concatenate([a["theta0"], b["theta0"][n:]]),
             pixels_yx=np.concatenate([a["pixels_yx"], b["pixels_yx"]]),
             pose_ids=np.concatenate([a["pose_ids"], b["pose_ids"]]),
             point_ids=np.concatenate([a["point_ids"], b["point_ids"] + a["M"]]))
    s["O"] = len(s["pose_ids"])
    return s


def _worker(rank, world, port, q, scene, die):
    sys.path.insert(0, ROOT)
    import datetime
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    torch.cuda.set_device(0)                                   # both ranks on GPU 0
    import slam_jl_amd  # noqa: F401
    from slam_jl_amd import sharded_ba, synthetic as syn
    s = _scene(syn, scene)
    if die and rank == 1:
        dist.barrier()
        os._exit(17)                                           # a rank that disappears before its first collective of the solve
    if die:
        dist.barrier()
    th, ol, st = sharded_ba.sharded_bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"],
                                                      host_paced=True)
    q.put((rank, th, ol, st))
    dist.barrier()
    dist.destroy_process_group()


def _run_two(scene, die=False, deadline=240):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, scene, die)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    res = []
    if not die:
        res = sorted([q.get(timeout=deadline) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(max(1.0, deadline - (time.time() - t0)))
    alive = [p.is_alive() for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()                                           # exactly the processes started above
    return res, [p.exitcode for p in procs], alive, time.time() - t0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("scene", ["P20", "P50", "unequal"])
def test_two_processes_on_one_gpu_equal_the_single_gpu_solve(slam, syn, scene):
    res, codes, alive, _ = _run_two(scene)
    assert codes == [0, 0] and not any(alive), (codes, alive)
    (_, t0, o0, s0), (_, t1, o1, s1) = res
    assert np.array_equal(t0, t1) and np.array_equal(o0, o1)                  # every rank returns the same full result
    s = _scene(syn, scene)
    assert s0["world_size"] == 2 and s0["points_local"] + s1["points_local"] == s["M"]
    if scene == "unequal":
        assert s0["obs_local"] == s1["obs_local"] == 6000                     # the split falls between the two kinds of points
    ref = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(ref, s["cam"])
    assert s0["iters_pass1"] == ref.stats["iters_pass1"] and s0["iters_pass2"] == ref.stats["iters_pass2"]
    assert np.array_equal(o0, ref.outliers) and s0["n_outliers"] == ref.stats["n_outliers"]
    assert abs(s0["ssr_final"] - ref.stats["ssr_final"]) <= 1e-8 * ref.stats["ssr_final"]
    assert np.abs(t0 - ref.theta).max() <= 1e-6


@pytest.mark.timeout(300)
def test_a_rank_that_dies_takes_the_survivor_out_of_its_collective():
    """rank 1 exits (os._exit) right before the solve; rank 0 must not wait for it for ever: its collective fails (gloo notices the closed
    connection, at the latest the 60 s process-group timeout) and the process exits non-zero"""
    _, codes, alive, dt = _run_two("P20", die=True, deadline=150)
    assert not any(alive), "the surviving rank is still waiting in a collective"
    assert codes[1] == 17 and codes[0] not in (0, None), codes
    assert dt < 150
