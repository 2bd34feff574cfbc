"""Where a single-stream frame's wall time goes (bench.GpuBackend / Stream, builds_in_flight = 3): python scripts/prof_single_loop.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
import bench
wl = bench.make_workload(slam, syn, "kitti05_1000", seed=0, streams=1)
dev = torch.device("cuda", 0)
ld = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in wl["left"]]
rd = [torch.from_numpy(np.ascontiguousarray(im.T)).to(dev) for im in wl["right"]]
torch.cuda.synchronize()
ahead = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c = [slam.Context(0) for _ in range(3 + ahead - 1)]
share = len(sys.argv) > 2 and sys.argv[2] == "share"          # the right builds share a left-build stream (one stream fewer: 4 hardware queues)
be = bench.GpuBackend(slam, c[0], c[1], c[1] if share else c[2], wl["H"], wl["W"], ld, rd, wl["params"], wl["extractor"], ahead=ahead, extra_build_ctx=(c[2:] if share else c[3:]))
acc = {}
kind = ["nf"]
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); key = name + ":" + ("stereo" if (name.startswith("match") and a and a[0]) else kind[0])
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0; return r
    return w
_bf = be.begin_frame
def bf(f_cur, upcoming, kf):
    kind[0] = "kf" if kf else "nf"
    return _bf(f_cur, upcoming, kf)
be.begin_frame = timed("begin_frame", bf); be.match = timed("match", be.match); be.detect = timed("detect", be.detect)
st = bench.Stream(be, wl["flows"], wl["disparity"], seed=0)
seq = bench.frame_sequence(400)
be.prime(seq[0])
for i in range(20): st.step(seq[i], seq[i + 1], seq[i + 2:i + 8])
be.drain(); acc.clear()
N = 200
t0 = time.perf_counter()
for i in range(20, 20 + N): st.step(seq[i], seq[i + 1], seq[i + 2:i + 8])
be.drain(); tot = time.perf_counter() - t0
nkf = N // 5; nnf = N - nkf
print(f"ahead {ahead}: {tot / N * 1e6:.1f} us per frame; per call of its kind (us): " + "; ".join(f"{k} {v / (nkf if (':kf' in k or ':stereo' in k) else nnf) * 1e6:.1f}" for k, v in sorted(acc.items())) + f"; python rest per frame {(tot - sum(acc.values())) / N * 1e6:.1f}")
