"""Does chunking the 64-image build help (intermediates of a chunk re-read while still in the 256 MB Infinity Cache)?
python scripts/prof_pyr_chunks.py : one S=64 batch vs C chunks of 64/C images on K contexts, 8-bit frames, graph replays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
H, W = syn.SHAPES["kitti05"]
left, right, flows = syn.stereo_stream("kitti05", 4, seed=0)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(np.round(im * 255).astype(np.uint8).T)).to(dev) for im in left]
torch.cuda.synchronize()
reps = 20
def run(C, K):
    S = 64 // C
    ctxs = [slam.Context(0) for _ in range(K)]
    pbs = [slam.PyramidBatch((H, W), levels=3, S=S, ctx=ctxs[c % K]) for c in range(C)]
    ptrs = [[ld[(c * S + s) % len(ld)].data_ptr() for s in range(S)] for c in range(C)]
    for r in range(2):
        for c in range(C): pbs[c].update_(ptrs[c], sync=False, ctx=ctxs[c % K], u8=True)
    for cc in ctxs: cc.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for c in range(C): pbs[c].update_(ptrs[c], sync=False, ctx=ctxs[c % K], u8=True)
    for cc in ctxs: cc.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e6
    for pb in pbs:
        for p in pb.pyramids: p.close()
    for cc in ctxs: cc.close()
    return dt
for C, K in ((1, 1), (2, 2), (4, 2), (4, 4), (8, 2), (8, 4), (8, 8), (16, 4)):
    print(f"{C} chunk(s) of {64 // C} images on {K} context(s): {run(C, K):.0f} us per 64 images")
