#!/usr/bin/env python3
"""Regenerates tests/golden/pose_v1.npz: regression vectors for the triangulation, P3P-RANSAC and five-point-RANSAC
entry points.  Like hotpath_v1.npz they are produced by the CPU oracle, not by the Julia
reference (no Julia here, no goldens in the reference, RecoverPose un-vendored): they freeze the oracle's behaviour,
which tests/test_oracle_{triangulation,p3p,5pt}.py pin against numpy and ground truth.

    python tests/golden/make_golden_pose.py
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


def main():
    out = {}
    t = syn.triangulation_scene(n=120, seed=21, noise_px=0.4, n_behind=6, n_gross=8)
    P1, P2 = projection_matrices(t["cam"], t["cam"], t["T21"])
    xyz, st = orc.triangulate(P1, P2, t["T21"], t["cam"], t["cam"], t["px1"], t["px2"], 3.0)
    out.update(tri_cam=np.array(t["cam"]), tri_T21=t["T21"], tri_px1=t["px1"], tri_px2=t["px2"], tri_xyz=xyz, tri_status=st)
    p = syn.p3p_scene(n=150, seed=22, noise_px=0.3, outlier_frac=0.2, iters=48)
    cnt, KP, Rt, inl, err, bi = orc.p3p_ransac(p["pts3d"], p["px_xy"], p["pdn"], p["K"], 3.0, p["samples"])
    out.update(p3p_pts=p["pts3d"], p3p_px=p["px_xy"], p3p_pdn=p["pdn"], p3p_K=p["K"], p3p_samples=p["samples"],
               p3p_n=cnt, p3p_KP=KP, p3p_Rt=Rt, p3p_inliers=inl, p3p_error=err, p3p_best=bi)
    f = syn.five_point_scene(n=150, seed=23, noise_px=0.3, outlier_frac=0.2, iters=24)
    cnt, E, P, inl, err, bi = orc.five_point_ransac(f["px1"], f["px2"], f["pd1"], f["pd2"], f["K"], f["K"], 3.0, f["samples"])
    out.update(fp_px1=f["px1"], fp_px2=f["px2"], fp_pd1=f["pd1"], fp_pd2=f["pd2"], fp_K=f["K"], fp_samples=f["samples"],
               fp_n=cnt, fp_E=E, fp_P=P, fp_inliers=inl, fp_error=err, fp_best=bi)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", "p3p inliers", out["p3p_n"], "five-point inliers", out["fp_n"])


if __name__ == "__main__":
    main()
