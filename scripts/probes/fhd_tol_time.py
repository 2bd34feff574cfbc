import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
H, W, S = 1080, 1920, 32
rng = np.random.default_rng(0)
u8 = (rng.random((4, W, H)) * 255).astype(np.uint8)
dev = torch.from_numpy(np.stack([u8[s % 4] for s in range(S)])).cuda(); torch.cuda.synchronize()
ptrs = [dev.data_ptr() + s * H * W for s in range(S)]
ctx = slam.Context(0)
pb = slam.PyramidBatch((H, W), levels=3, S=S, ctx=ctx)
for fast in (False, True):
    for _ in range(3): pb.update_(ptrs, u8=True, fast=fast, ctx=ctx)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(10): pb.update_(ptrs, u8=True, fast=fast, sync=False, ctx=ctx)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 10
    alg = S * 7 * 8 * sum((-(-H // 2**l)) * (-(-W // 2**l)) for l in range(4))
    print(f"FHD S={S} tol={fast}: {dt*1e6:.0f} us/build, frac {alg/dt/8e12:.3f}")
