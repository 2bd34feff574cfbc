"""GPU: bench.py's two single-stream backends must track identically -- five single-image builds in flight (GpuBackend) against the next
key-frame period built as ONE batched launch set (GpuPeriodBackend: `single_stream.value` since the end of round 3).  The batch kernels are
the single-image recurrences with grid.z = S, so every plane, and with it every keypoint list, is the same bit for bit."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_period_batched_builds_track_like_single_image_builds(slam, syn):
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); sys.modules["bench_mod"] = bench; spec.loader.exec_module(bench)
    H, W = 150, 200
    left, right, flows = syn.stereo_stream((H, W), 8, seed=3, step=(1.1, -1.7), disparity=6.0)
    params = slam.Params(stereo=True, max_nb_keypoints=120)
    e = slam.Extractor.from_params(params, slam.Camera(*syn.KITTI_CAM, height=H, width=W))
    ld = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in left]
    rd = [torch.from_numpy(np.ascontiguousarray(im.T)).cuda() for im in right]
    torch.cuda.synchronize()
    seq = bench.frame_sequence(60)
    AH = 2 * bench.KF_EVERY
    lists = []
    for period in (False, True):
        ctxs = [slam.Context(0) for _ in range(7)]
        if period:
            be = bench.GpuPeriodBackend(slam, ctxs[0], ctxs[1], H, W, ld, rd, params, e, bench.KF_EVERY)
        else:
            be = bench.GpuBackend(slam, ctxs[0], ctxs[1], ctxs[2], H, W, ld, rd, params, e, ahead=5, extra_build_ctx=ctxs[3:])
        st = bench.Stream(be, flows, 6.0, seed=0)
        be.prime(seq[0])
        snaps = []
        for i in range(32):
            st.step(seq[i], seq[i + 1], seq[i + 2:i + 2 + AH])
            snaps.append((st.kp.copy(), st.is3d.copy()))
        be.drain(); be.close()
        for c in ctxs:
            c.close()
        lists.append(snaps)
    assert len(lists[0][-1][0]) > 40                                            # something is being tracked
    for i, (a, b) in enumerate(zip(*lists)):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), f"keypoint lists differ after frame {i + 1}"
