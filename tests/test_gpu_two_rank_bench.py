"""GPU: bench.py's N > 1 path on a 1-GPU box -- `python bench.py --gpus 2` starts two rank processes itself (SLAM_BENCH_ONE_GPU: both on
GPU 0, collectives over gloo; on a multi-GPU node the same code runs one rank per GPU over RCCL, which the build environment cannot
exercise).  ONE line with n_gpus = 2, value = all frames over the slowest rank's time, the per-rank values listed; a rank that dies makes
the whole job exit non-zero instead of hanging."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, timeout=600):
    env = dict(os.environ, SLAM_BENCH_ONE_GPU="1", SLAM_BENCH_SPAWN_TIMEOUT_S="500", **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--only", "headline", "--steps", "4", "--warmup", "2", "--streams", "8"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=timeout, cwd=ROOT)


def test_two_ranks_print_one_line_with_the_sum():
    r = _run({})
    assert r.returncode == 0, r.stdout[-500:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["streams_per_gpu"] == 8
    pr = j["per_rank_values"]
    assert len(pr) == 2 and all(v > 0 for v in pr)
    # value = all frames / the slowest rank's seconds = 2 x the slowest rank's own rate <= the sum of the ranks' own rates
    assert j["value"] <= sum(pr) * 1.0001 and j["value"] >= 2 * min(pr) * 0.98, (j["value"], pr)


def test_a_dead_rank_ends_the_job_with_an_error():
    r = _run({"SLAM_BENCH_KILL_RANK": "1"}, timeout=300)
    assert r.returncode != 0
