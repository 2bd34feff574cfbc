#!/usr/bin/env python3
"""Probe for the monocular pose loop of bench.py (configs[3], benchlib/lockstep.py pose=True, stereo=False): per frame step the
translation error of every stream, the number of map points, and the error of the map points against the scene plane.

    python scripts/probes/mono_debug.py [S] [periods]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import slam_jl_amd as slam  # noqa: E402
from slam_jl_amd import synthetic as syn  # noqa: E402
from benchlib.lockstep import run_lockstep_kpset, make_workload  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
periods = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = make_workload(slam, syn, "euroc_mono", seed=0, streams=S)
camt = wl["camt"]
Z = 30.0
dev = torch.device("cuda", 0)
rows = []


def diag(i, kf, pst, ks, ctx, off_now):
    if pst["ref"] is None:
        return
    off = off_now - pst["ref"]
    want = np.stack([off[:, 1] * Z / camt[0], off[:, 0] * Z / camt[1], np.zeros(S)], axis=1)
    terr = np.abs(pst["Tcw"][:, :3, 3] - want).max(axis=1)
    rot = np.abs(pst["Tcw"][:, :3, :3] - np.eye(3)).max(axis=(1, 2))
    n3, perr, zmin, zmax, nbad = [], [], [], [], []
    for s in range(S):
        d = ks.download(s, ctx=ctx)
        m = d["is_3d"]
        n3.append(int(m.sum()))
        if m.any():
            yx = d["yx"][m]
            ref_px = yx - off[s]
            Xw = np.stack([(ref_px[:, 1] - camt[2]) / camt[0] * Z, (ref_px[:, 0] - camt[3]) / camt[1] * Z, np.full(len(yx), Z)], axis=1)
            e = np.abs(d["xyz"][m] - Xw).max(axis=1)
            perr.append(float(np.median(e))); zmin.append(float(d["xyz"][m][:, 2].min())); zmax.append(float(d["xyz"][m][:, 2].max()))
            nbad.append(int((e > 3.0).sum()))
        else:
            perr.append(0.0); zmin.append(0.0); zmax.append(0.0); nbad.append(0)
    w = int(np.argmax(terr))
    print(f"i={i:3d} kf={int(kf)} terr max {terr.max():9.4f} (stream {w}, rot {rot[w]:.2e})  n3d {min(n3)}..{max(n3)}  "
          f"map err median {max(perr):8.3f}  z {min(zmin):10.2f}..{max(zmax):10.2f}  points > 3 m off: {max(nbad)}", flush=True)
    rows.append((i, terr.copy()))


r = run_lockstep_kpset(slam, torch, 0, wl, periods, 2, 1, None, dev, "host_u8", pose=True, diag=diag)
print(r["pose"])
