import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn, sharded_ba
import bench
dev = torch.device("cuda", 0)
H, W = syn.SHAPES["kitti05"]
params = slam.Params(stereo=True, max_nb_keypoints=1000)
cam = slam.Camera(*syn.KITTI_CAM, height=H, width=W)
ex = slam.Extractor.from_params(params, cam)
left, right, flows = syn.stereo_stream("kitti05", 8, seed=0, disparity=12.4)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "kpset"):
    r = bench.run_lockstep_kpset(slam, torch, 0, 8, 12, 6, H, W, left, right, flows, 12.4, params, ex, 1, None, dev, "host_u8")
    print("kpset leg ok", r["value"])
s2 = syn.ba_scene(P=20, M=2000, seed=7)
try:
    out = sharded_ba.sharded_bundle_adjustment(s2["cam"], s2["theta0"], s2["theta_const"], s2["pixels_yx"], s2["pose_ids"], s2["point_ids"])
    print("sharded ok", out[2]["ssr_final"])
except Exception:
    traceback.print_exc()
