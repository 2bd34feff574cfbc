#!/usr/bin/env python3
"""Regenerates tests/golden/hotpath_v1.npz.

These are REGRESSION vectors produced by the CPU oracle (oracle/), not outputs
of the Julia reference: pxl-th/SLAM.jl ships no tests or golden vectors
 and Julia is not installed here, so reference-pinned fixtures cannot
exist.  They freeze the oracle's behaviour (which tests/test_oracle_*.py pin
against independent numpy/scipy restatements) so that any later change to the
oracle or to the HIP path that alters results is caught.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import slam_jl_amd  # noqa: E402,F401
from slam_jl_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    H, W = 70, 105
    L, R, flows = syn.stereo_stream((H, W), 2, seed=42, step=(0.8, -1.1), disparity=3.2)
    img0 = np.round(L[0] * 255) / 255          # 8-bit quantised like a decoded PNG; stored as uint8
    img1 = np.round(L[1] * 255) / 255
    cur = np.array([[20.0, 30.0], [50.5, 80.5], [1.0, 1.0]])
    kp0 = orc.detect(img0, np.zeros((0, 2)), max_points=60)
    kp1 = orc.detect(img0, cur, max_points=60)
    p0 = orc.pyr_build(img0, 2, 1.0, 1); p1 = orc.pyr_build(img1, 2, 1.0, 1); pc = orc.pyr_build(img0, 2, 1.0, 0)
    out, st = orc.fb_tracking(p0, p1, kp0.astype(float), sum_order=1, pyramid_levels=2)
    s = syn.ba_scene(P=4, M=40, seed=9, obs_per_point=3)
    th, ol, stats = orc.bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"], solver=1)
    pat = np.clip(np.rint(np.random.default_rng(123).normal(0, 9 / 5, (256, 4))), -4, 4).astype(np.int32)
    bits, brc = orc.describe(img0, kp0, pat)
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "hotpath_v1.npz"),
        img0_u8=np.round(img0 * 255).astype(np.uint8), img1_u8=np.round(img1 * 255).astype(np.uint8),
        cur=cur, kp_nomask=kp0, kp_mask=kp1,
        upd_Iy_l1=p0.plane("Iy", 1), upd_Iyx_l2=p0.plane("Iyx", 2), upd_layer_l2=p0.plane("layers", 2),
        ctor_Ixx_l1=pc.plane("Ixx", 1), ctor_layer_l1=pc.plane("layers", 1),
        lk_out=out, lk_status=st,
        ba_theta0=s["theta0"], ba_const=s["theta_const"], ba_pixels=s["pixels_yx"], ba_pose_ids=s["pose_ids"],
        ba_point_ids=s["point_ids"], ba_cam=np.array(s["cam"]), ba_theta=th, ba_outliers=ol,
        ba_ssr=np.array([stats["ssr_init"], stats["ssr_pass1"], stats["ssr_final"]]),
        brief_pattern=pat, brief_bits=bits, brief_rc=brc)
    print("wrote hotpath_v1.npz:", len(kp0), "kps,", int(st.sum()), "tracked, BA ssr", stats["ssr_final"])


if __name__ == "__main__":
    main()
