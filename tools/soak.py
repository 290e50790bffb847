"""Soak: N consecutive GAMG p-solves (the bench step) on one mesh; per-solve wall times, their spread, engine fallbacks.
Looks for the rare crawling launch (watchdog) - every solve slower than 1.5 x the median is listed.
python tools/soak.py [box:216 | octree:14:6:7] [solves=300]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
import torch
from openfoam_amd import capi, cases, octree

spec = sys.argv[1] if len(sys.argv) > 1 else "box:216"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
f = spec.split(":")
if f[0] == "box":
    p = cases.box3d(int(f[1]))
else:
    q = int(f[1])
    p = octree.problem(base=(5 * q, 2 * q, 2 * q), surface_levels=(int(f[2]), int(f[3])))
    p.pop("cellLevel")
    order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
    nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
    p = cases.renumbered(p, order, fmap, flip, nl, nu)
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
addr = capi.Addressing(ctx, p["nCells"], p["lowerAddr"], p["upperAddr"], p.get("faceWeights"))
mat = capi.Matrix(addr)
d_diag, d_upper, d_source = (torch.from_numpy(p[k]).to(dev) for k in ("diag", "upper", "source"))
d_psi = torch.zeros(p["nCells"], dtype=torch.float64, device=dev)
controls = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10, mergeLevels=1,
                cacheAgglomeration=1, nPreSweeps=0, nPostSweeps=2, nFinestSweeps=2, tolerance=1e-7, relTol=0.01)
times, its, res = [], [], []
for i in range(N + 1):
    d_psi.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mat.set_coeffs(d_diag, d_upper)
    _, perf = mat.solve(d_psi, d_source, **controls)
    ctx.sync()
    dt = time.perf_counter() - t0
    if i:   # (the first solve builds the hierarchy)
        times.append(dt * 1e3); its.append(perf["nIterations"]); res.append(perf["finalResidual"])
t = np.array(times)
med = np.median(t)
print("%s: %d cells, %d solves of %d V-cycles: ms per solve min %.3f  median %.3f  mean %.3f  p99 %.3f  max %.3f; engine fallbacks %d; "
      "final residuals identical: %s" % (spec, p["nCells"], N, its[0], t.min(), med, t.mean(), np.percentile(t, 99), t.max(),
                                         ctx.fallback_count(), len(set(res)) == 1 and len(set(its)) == 1))
slow = [(i + 1, round(x, 3)) for i, x in enumerate(t) if x > 1.5 * med]
print("solves slower than 1.5 x the median:", slow if slow else "none")
