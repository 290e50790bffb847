"""GAMG p-solve on the irregular (box + diagonals, bandCompression-renumbered) stand-in: time, fallbacks, per-level info.
python tools/irregular_gamg_probe.py [n=100]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.join(HERE, ".."))
import __graft_entry__ as entry
entry.load_package()
import torch, numpy as np
from openfoam_amd import capi, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
p = cases.irregular_box(n)
order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
p = cases.renumbered(p, order, fmap, flip, nl, nu)
ctx = capi.Context(0)
addr = capi.Addressing(ctx, p["nCells"], p["lowerAddr"], p["upperAddr"], p["faceWeights"])
mat = capi.Matrix(addr)
dev = torch.device("cuda", 0)
d_diag, d_upper, d_source = (torch.from_numpy(p[k]).to(dev) for k in ("diag", "upper", "source"))
d_psi = torch.zeros(p["nCells"], dtype=torch.float64, device=dev)
kw = dict(solver="GAMG", tolerance=1e-7, relTol=0.01, smoother="GaussSeidel", nPreSweeps=0, nPostSweeps=2, nFinestSweeps=2,
          cacheAgglomeration=1, agglomerator="faceAreaPair", nCellsInCoarsestLevel=10, mergeLevels=1)
def step():
    d_psi.zero_(); torch.cuda.synchronize()
    mat.set_coeffs(d_diag, d_upper)
    return mat.solve(d_psi, d_source, history=True, **kw)[1]
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); its = 0
for _ in range(3): its += step()["nIterations"]
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("irregular %d^3: %d V-cycles in %.1f ms -> %.2f V-cycles/s; fallbacks %d" % (n, its, dt * 1e3, its / dt, ctx.fallback_count()))
for lv, row in enumerate(mat.gamg_level_sizes(**kw)):
    print("  level", lv, row)
