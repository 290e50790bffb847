"""The GAMG p-solve of bench.py alone (no PCG leg, no CPU leg) for rocprofv3 traces:
python tools/gamg_profile.py [n=216] [solves=4] [mode: box|renumbered|random]"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import __graft_entry__ as entry
entry.load_package()
import torch
from openfoam_amd import capi, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = cases.box3d(n)
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
# steady-state marker for tools/trace_steady.py: a second context runs the placement census kernel once; nothing
# after it belongs to set-up (which uploads its tables through hundreds of staged copyBuffer blits)
step(); torch.cuda.synchronize()
marker_ctx = capi.Context(0)
torch.cuda.synchronize()
print("MARK warm-up done", flush=True)
t0 = time.perf_counter(); its = 0
for _ in range(solves):
    its += step()["nIterations"]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("GAMG only: %d solves, %d V-cycles, %.3f ms per solve, %.1f V-cycles/s" % (solves, its, 1e3 * dt / solves, its / dt))
