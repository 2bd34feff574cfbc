import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import slam_jl_amd as s
from slam_jl_amd import _lib as L, synthetic as syn
from oracle import oracle as orc
ctx = s.default_context(0)
sc = syn.ba_scene(P=5, M=300, seed=0)
P, M, O = sc['P'], sc['M'], sc['O']
h = C.c_void_p()
th = sc['theta0'].copy(); tc = sc['theta_const']; px = sc['pixels_yx']; pi = sc['pose_ids']; li = sc['point_ids']
ctx.check(ctx.lib.slam_ba_create(ctx.h, *sc['cam'], P, M, O, L.ptr(th), L.ptr(tc, L.u8p), L.ptr(px), L.ptr(pi, L.i64p), L.ptr(li, L.i64p), C.byref(h)))
n = 6 * P
rl = ctx.lib.slam_ba_reduce_len(P)
red = torch.zeros(rl, dtype=torch.float64, device='cuda')
ctx.check(ctx.lib.slam_ba_build(ctx.h, h, 0, 0.1, C.c_void_p(red.data_ptr())))
r = red.cpu().numpy()
S = r[:n*n].reshape(n, n, order='F'); g = r[n*n:n*n+n]; ud = r[n*n+n:n*n+2*n]; ssr = r[n*n+2*n]
So, go, udo, ssro = orc.ba_reduced_system(sc['cam'], th, tc, px, pi, li, np.zeros(O, np.uint8), 0, 0.1, 0, M)
print("ssr", ssr, ssro)
print("S diff", np.abs(S - So).max(), np.abs(So).max())
print("g diff", np.abs(g - go).max(), np.abs(go).max())
print("ud diff", np.abs(ud - udo).max(), np.abs(udo).max())
print(np.linalg.eigvalsh(So + np.diag(np.clip(udo, 1e-6, 1e32) * 0.1))[:3])
print(np.linalg.eigvalsh(S + np.diag(np.clip(ud, 1e-6, 1e32) * 0.1))[:3])
tr = torch.zeros(4, dtype=torch.float64, device='cuda')
ctx.check(ctx.lib.slam_ba_solve(ctx.h, h, C.c_void_p(red.data_ptr()), 0.1, C.c_void_p(tr.data_ptr())))
print("trial", tr.cpu().numpy())
