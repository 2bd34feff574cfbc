"""Serial-launch batched pyramid updates (S images, 370x1226) for rocprofv3 --pmc passes: python3 scripts/pmc_probe_batch.py [S] [fast|u8|u8tol]
(u8: 8-bit frames through the fused ingest of k_cols_fused -- the headline's build)"""
import os, sys
os.environ["SLAMHIP_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fast = len(sys.argv) > 2 and sys.argv[2] in ("fast", "u8tol")      # u8tol: 8-bit ingest + the tolerance-mode batch kernels (mode 3)
u8 = len(sys.argv) > 2 and sys.argv[2] in ("u8", "u8tol")
H, W = 370, 1226
L, R, flows = syn.stereo_stream((H, W), 2, seed=0)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray((np.round(im * 255).astype(np.uint8) if u8 else im).T)).to(dev) for im in L]
torch.cuda.synchronize()
ctx = slam.Context(0)
pb = slam.PyramidBatch((H, W), levels=3, S=S, ctx=ctx)
for i in range(6):
    pb.update_([ld[(i + s) % 2].data_ptr() for s in range(S)], sync=True, fast=fast, ctx=ctx, u8=u8)
