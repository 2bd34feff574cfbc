"""The headline loop alone (bench.run_lockstep_kpset, 100 steps): python scripts/prof_headline.py [host_u8|dev_f64|host_f64]
env: S_ streams per GPU (default 32), SLAM_BENCH_TRACK_PRIO / SLAM_BENCH_PYR_PRIO / SLAM_BENCH_CU_SPLIT as in bench.py; prints frames/s, ms per step,
the in-pipeline build time and the host's wait per step.  Under rocprofv3 --kernel-trace + scripts/kernel_timeline.py: the step's timeline."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
import bench
dev = torch.device("cuda", 0)
S_ = int(os.environ.get("S_", "64"))
wl = bench.make_workload(slam, syn, os.environ.get("WL_", "kitti05_1000"), seed=0, streams=S_)
r = bench.run_lockstep_kpset(slam, torch, 0, wl, int(os.environ.get("PERIODS_", "20")), 2, 1, None, dev, sys.argv[1] if len(sys.argv) > 1 else "host_u8")
print("lib", os.environ.get("SLAMHIP_LIB", "default"), "value", round(r["value"]), "ms/frame-step", round(r["ms_per_frame_of_S_streams"], 3), "build ms", round(r["pyramid_build_ms"]["mean"] or 0, 3),
      "host wait ms/frame", round(r["host_wait_ms_per_frame"], 3), "tracked", r["tracked_kpts_per_frame"])
