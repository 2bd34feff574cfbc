#!/usr/bin/env python3
"""Regenerates tests/golden/frontend_v1.npz: inputs + oracle outputs for the call protocol of optical_flow_matching!
(map_manager.jl:451-564 temporal, and the stereo form with maybe_stereo_update! :579-590) followed by triangulate_stereo!
(mapper.jl:142-183).  Like hotpath_v1.npz these are REGRESSION vectors produced by the CPU oracle; tests/golden/make_golden_julia.jl
feeds the same inputs to the real SLAM.jl (MapManager + Frame built from these arrays) and writes julia_frontend_v1.npz, which
tests/test_golden_julia.py compares with -- that is what pins this part of the oracle.

    python tests/golden/make_golden_frontend.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import slam_jl_amd  # noqa: E402,F401
from slam_jl_amd import synthetic as syn  # noqa: E402
from slam_jl_amd.triangulation import projection_matrices  # noqa: E402
from oracle import oracle as orc  # noqa: E402

H, W, DISP, BASELINE = 120, 160, 6.3, 0.54
CAM = (180.0, 180.0, 80.0, 60.0)            # fx, fy, cx, cy of a 120 x 160 pinhole camera, no distortion


def main():
    L, R, flows = syn.stereo_stream((H, W), 2, seed=77, step=(0.9, -1.2), disparity=DISP)
    q = lambda im: np.round(im * 255).astype(np.uint8)
    f = lambda u8: np.asfortranarray(u8.astype(np.float64) / 255.0)
    l0, l1, r1 = q(L[0]), q(L[1]), q(R[1])
    rng = np.random.default_rng(5)
    kp = orc.detect(f(l0), np.zeros((0, 2)), max_points=80).astype(np.float64)
    n = len(kp)
    is3d = rng.random(n) < 0.6
    proj = kp + np.array(flows[1]) + rng.normal(0, 0.3, kp.shape)
    proj[::9] += 25.0                                   # bad priors: the 3-D attempt fails, the keypoint joins the 2-D pass
    is3d[1] = True; proj[1] = (H + 3.0, 10.0)           # projection outside the image: skipped (temporal)
    p0, p1, pr = (orc.pyr_build(f(x), 3, 1.0, 1) for x in (l0, l1, r1))
    t = orc.optical_flow_matching(p0, p1, kp, is3d, proj, (H, W), sum_order=0)
    keep = ~t["removed"]
    kp1, is3d1 = t["new_pixels"][keep], is3d[keep]
    # stereo: left frame 1 -> right frame 1; priors of the 3-D keypoints = pixel shifted by the disparity (+ noise)
    sproj = kp1 + np.array([0.0, -DISP]) + rng.normal(0, 0.3, kp1.shape)
    s = orc.optical_flow_matching(p1, pr, kp1, is3d1, sproj, (H, W), stereo=True, undistorted_left=kp1, right_cam=CAM, sum_order=0)
    skeep = ~s["removed"]
    kp2, is3d2, up, syx = kp1[skeep], is3d1[skeep], s["updated"][skeep], s["new_pixels"][skeep]
    T21 = np.eye(4); T21[0, 3] = -BASELINE
    P1, P2 = projection_matrices(CAM, CAM, T21)
    cand = np.flatnonzero(up & ~is3d2)
    xyz, ok = orc.triangulate(P1, P2, T21, CAM, CAM, kp2[cand], syx[cand], 3.0)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend_v1.npz")
    np.savez_compressed(path, l0_u8=l0, l1_u8=l1, r1_u8=r1, cam=np.array(CAM), baseline=np.array([BASELINE]), disparity=np.array([DISP]),
                        kp=kp, is3d=is3d, proj=proj,
                        t_new=t["new_pixels"], t_updated=t["updated"], t_removed=t["removed"],
                        s_proj=sproj, s_new=s["new_pixels"], s_updated=s["updated"], s_removed=s["removed"],
                        tri_cand=cand, tri_xyz=xyz, tri_ok=ok)
    print("wrote", path, ":", n, "keypoints,", int(t["updated"].sum()), "tracked,", int(up.sum()), "stereo matches,", int(ok.sum()), "of", len(cand), "triangulated")


if __name__ == "__main__":
    main()
