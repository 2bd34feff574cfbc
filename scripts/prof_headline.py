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
H, W = syn.SHAPES["kitti05"]
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream("kitti05", 8, seed=0, disparity=12.4)
S_ = int(os.environ.get("S_", "32"))
r = bench.run_lockstep_kpset(slam, torch, 0, S_, 100, 10, H, W, left, right, flows, 12.4, params, ex, 1, None, dev, sys.argv[1] if len(sys.argv) > 1 else "host_u8")
print("split", os.environ.get("SLAM_BENCH_CU_SPLIT"), "value", round(r["value"]), "ms/step", round(r["ms_per_step_of_S_frames"], 3), "build ms", round(r["pyramid_build_ms"]["mean"] or 0, 3), "host wait ms/step", round(r["host_wait_ms_per_step"], 3))
