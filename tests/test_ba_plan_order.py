"""The pose order slam_local_ba solves in (`slam_ba_plan_order`, host work only: runs without a GPU).  A window of consecutive
key-frames is block-banded as it is; a window with loop closures (src/map_manager.jl:300-449: old map points re-associated with the
newest key-frames) is a ring, and is banded once folded."""
import numpy as np
import pytest

import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn


def _cache(s):
    return slam.LocalBACache(s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])


def _halfband_in(s, order):
    """half-bandwidth of the free poses' covisibility in the given order -- numpy restatement of the definition"""
    P = s["P"]
    new_of = np.empty(P, dtype=np.int64); new_of[np.asarray(order)] = np.arange(P)
    free = np.asarray(s["theta_const"])[s["pose_ids"] - 1] == 0
    pid, pos = s["point_ids"][free], new_of[s["pose_ids"][free] - 1]
    lo = np.full(s["M"] + 1, 1 << 30); hi = np.full(s["M"] + 1, -1)
    np.minimum.at(lo, pid, pos); np.maximum.at(hi, pid, pos)
    seen = hi >= 0
    return int((hi[seen] - lo[seen]).max())


def test_a_chain_keeps_the_callers_order():
    s = syn.ba_scene(P=30, M=600, seed=1)
    order, hb, reordered = slam.ba_plan_order(_cache(s))
    assert not reordered and hb == 9 and np.array_equal(order, np.arange(30))
    s = syn.ba_scene(P=26, M=300, seed=2, obs_per_point=24)           # a dense band of 24: no order helps, the general path stays
    order, hb, reordered = slam.ba_plan_order(_cache(s))
    assert not reordered and hb == 23 and np.array_equal(order, np.arange(26))


@pytest.mark.parametrize("P,k_loop,n_const", [(50, 5, 1), (30, 5, 1), (40, 3, 2), (60, 8, 1), (23, 5, 1)])
def test_a_ring_is_folded_into_the_band(P, k_loop, n_const):
    s = syn.ba_scene_loop(P=P, M=20 * P, seed=P, n_loop=50, k_loop=k_loop, n_const=n_const)
    assert syn.ba_halfband(s) > 20
    order, hb, reordered = slam.ba_plan_order(_cache(s))
    assert reordered and hb <= 20
    assert sorted(order.tolist()) == list(range(P))                                   # a permutation
    nc = int(np.asarray(s["theta_const"]).sum())
    assert np.asarray(s["theta_const"])[order[:nc]].all()                             # constant poses first
    assert _halfband_in(s, order) == hb


def test_constant_poses_between_free_ones_do_not_count():
    """free poses 0, 3, 6, ... with constant ones between them and points seen by 8 consecutive FREE poses: span 21 in the caller's
    order, 7 with the constant poses moved out of the way"""
    P = 60
    s = syn.ba_scene(P=P, M=900, seed=4, obs_per_point=22)
    const = np.ones(P, dtype=np.uint8); const[::3] = 0
    s["theta_const"] = const
    assert syn.ba_halfband(s) == 21
    order, hb, reordered = slam.ba_plan_order(_cache(s))
    assert reordered and hb == 7 and _halfband_in(s, order) == 7


def test_bad_ids_are_refused():
    s = syn.ba_scene(P=6, M=20, seed=5, obs_per_point=4)
    s["pose_ids"] = s["pose_ids"].copy(); s["pose_ids"][3] = 7
    with pytest.raises(slam.SlamHipError):
        slam.ba_plan_order(_cache(s))


def test_random_covisibility_graphs_property():
    """slam_ba_plan_order on random observation structures (no geometry needed: it only reads the ids and the constant flags) -- chains cut
    into pieces, rings, stars, random sparse graphs, isolated poses: the order is a permutation with the constant poses first, the
    half-bandwidth it reports is the one of that order, and a window is only reordered when that brings it into the banded solver's range."""
    rng = np.random.default_rng(0)
    n_reordered = 0
    for trial in range(300):
        P = int(rng.integers(3, 120)); M = int(rng.integers(1, 400))
        const = (rng.random(P) < rng.choice([0.0, 0.1, 0.5])).astype(np.uint8)
        kind = trial % 5
        pose_ids, point_ids = [], []
        for j in range(M):
            if kind == 0:   obs = (int(rng.integers(0, P)) + np.arange(int(rng.integers(2, 12)))) % P                  # ring
            elif kind == 1: a = int(rng.integers(0, P)); obs = np.arange(a, min(P, a + int(rng.integers(2, 12))))       # chain
            elif kind == 2: obs = np.unique(np.concatenate([[0], rng.integers(0, P, 2)]))                              # star around pose 0
            elif kind == 3: obs = np.unique(rng.integers(0, P, int(rng.integers(2, 6))))                               # random sparse
            else:                                                                                                      # two interleaved chains
                a = int(rng.integers(0, P // 2 + 1)); obs = 2 * np.arange(a, min(P // 2, a + 6)) + int(rng.integers(0, 2)); obs = obs[obs < P]
            if len(obs) == 0: obs = np.array([0])
            pose_ids += [int(o) + 1 for o in obs]; point_ids += [j + 1] * len(obs)
        s = dict(P=P, M=M, theta_const=const, pose_ids=np.asarray(pose_ids, np.int64), point_ids=np.asarray(point_ids, np.int64))
        cache = slam.LocalBACache(np.zeros(6 * P + 3 * M), const, np.zeros((len(pose_ids), 2)), s["pose_ids"], s["point_ids"])
        order, hb, reordered = slam.ba_plan_order(cache)
        assert sorted(order.tolist()) == list(range(P)), trial
        free = const[s["pose_ids"] - 1] == 0
        hb_id = syn.ba_halfband(s) if free.any() else 0
        hb_new = _halfband_in(s, order) if free.any() else 0
        assert hb == hb_new, (trial, hb, hb_new)
        if reordered:
            n_reordered += 1
            assert hb_id > 20 and hb <= 20, (trial, hb_id, hb)
            nc = int(const.sum())
            assert const[order[:nc]].all(), trial
        else:
            assert np.array_equal(order, np.arange(P)) and hb == hb_id, trial
    assert n_reordered > 20
