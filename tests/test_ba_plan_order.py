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
