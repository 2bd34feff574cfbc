import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import sharded_ba, synthetic as syn
s = syn.ba_scene(P=12, M=1500, seed=21)
for rep in range(3):
    th, ol, st = sharded_ba.sharded_bundle_adjustment(s["cam"], s["theta0"], s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    cache = slam.LocalBACache(s["theta0"].copy(), s["theta_const"], s["pixels_yx"], s["pose_ids"], s["point_ids"])
    slam.bundle_adjustment_(cache, s["cam"])
    print(st, cache.stats)
    print("outl equal", np.array_equal(ol, cache.outliers), "theta diff", np.abs(th - cache.theta).max())
