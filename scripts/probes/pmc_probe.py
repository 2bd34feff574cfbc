"""Runs 20 serial-launch pyramid updates (370x1226) for rocprofv3 --pmc passes (scratch tool)."""
import os, sys
os.environ["SLAMHIP_NO_GRAPH"] = "1"
sys.path.insert(0, '/root/repo')
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
H, W = 370, 1226
L, R, flows = syn.stereo_stream((H, W), 2, seed=0)
cur = slam.LKPyramid(shape=(H, W), levels=3)
for i in range(20):
    slam.update_(cur, L[i % 2])
