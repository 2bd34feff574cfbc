"""Batched pyramid build alone: python scripts/prof_pyr_batch.py [S] [reps]
prints the graph-replay time per build (hipEvents through slam_event), and the serial-launch time.
Env toggles of csrc/pyramid.hip (SLAMHIP_NO_COLS_FUSED, SLAMHIP_NO_ROWS_RESIZE, ...) select variants;
run under `rocprofv3 --kernel-trace --stats` for per-kernel numbers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
u8 = len(sys.argv) > 3 and sys.argv[3] == "u8"
fast = len(sys.argv) > 4 and sys.argv[4] == "tol"        # mode 3: the tolerance-mode batch kernels
H, W = syn.SHAPES[os.environ.get('SHAPE', 'kitti05')]
left, right, flows = syn.stereo_stream(os.environ.get('SHAPE', 'kitti05'), 8, seed=0, disparity=12.4)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray((np.round(im * 255).astype(np.uint8) if u8 else im).T)).to(dev) for im in left]
torch.cuda.synchronize()
ctx = slam.Context(0)
pb = slam.PyramidBatch((H, W), levels=3, S=S, ctx=ctx)
ptrs = [ld[s % len(ld)].data_ptr() for s in range(S)]
pb.update_(ptrs, sync=True, ctx=ctx, u8=u8, fast=fast)
pb.update_(ptrs, sync=True, ctx=ctx, u8=u8, fast=fast)
t0 = time.perf_counter()
for _ in range(reps):
    pb.update_(ptrs, sync=False, ctx=ctx, u8=u8, fast=fast)
ctx.synchronize()
g = (time.perf_counter() - t0) / reps * 1e6
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(reps):
    pb.update_(ptrs, sync=False, ctx=ctx, u8=u8, fast=fast)
ctx.synchronize()
ms, n = ctx.prof_get("pyr_update"); rms, rn = ctx.prof_get("k_iir_rows")
ctx.prof_enable(False)
alg = S * bench.pyramid_bytes(H, W, 3)
print(f"S={S} u8={u8} tol={fast} graph replay {g:.1f} us/build ({alg / g / 1e3:.1f} GB/s algorithmic, frac {alg / g / 1e3 / 8000:.3f}); "
      f"serial launches {ms / n * 1e3:.1f} us/build (frac {alg / (ms / n * 1e3) / 1e3 / 8000:.3f}); k_iir_rows {rms / rn * 1e3:.1f} us/launch")
