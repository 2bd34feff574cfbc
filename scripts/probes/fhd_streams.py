#!/usr/bin/env python3
"""configs[4] (1080 x 1920, 4000 keypoints) through the headline loop at several streams-per-GPU: python scripts/probes/fhd_streams.py S [tol]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
from benchlib.lockstep import run_lockstep_kpset, make_workload
S = int(sys.argv[1]); tol = len(sys.argv) > 2 and sys.argv[2] == "tol"
wl = make_workload(slam, syn, "fhd_4000", seed=0, streams=S)
if tol: wl["tolerance"] = True
r = run_lockstep_kpset(slam, torch, 0, wl, 5, 2, 1, None, torch.device("cuda", 0), "host_u8")
print(f"fhd_4000 S={S} tol={tol}: {r['value']:.0f} frames/s, build {r['pyramid_build_ms']['mean']:.2f} ms, hbm {r['hbm_in_use_gb']:.1f} GB, tracked {r['tracked_kpts_per_frame']}")
