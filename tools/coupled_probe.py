"""GPU probe: coupled PBiCCCG + DILU (type coupled, U-equation like) iteration rate at n^3, per-kernel split."""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import conftest  # noqa
from openfoam_amd import capi, cases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = cases.box3d(n, asym=True)
rng = np.random.RandomState(0)
psi, src = np.zeros((p["nCells"], 3)), rng.randn(p["nCells"], 3)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
for solver in ("PBiCCCG", "PBiCICG", "SmoothSolver"):
    for rep in range(2):
        x, perf = m.coupled_solve(psi, src, solver=solver, preconditioner="DILU", tolerance=1e-30, maxIter=iters - 1,
                                  nSweeps=1)
    print("%s n=%d^3 cells=%d iters=%d solve=%.4f s -> %.2f it/s  (%.3f ms/it)" % (
        solver, n, p["nCells"], perf["nIterations"], perf["solveSeconds"], perf["nIterations"] / perf["solveSeconds"],
        1e3 * perf["solveSeconds"] / perf["nIterations"]), flush=True)
# scalar PBiCG for comparison (one component)
x, perf = m.solve(np.zeros(p["nCells"]), src[:, 0].copy(), solver="PBiCG", preconditioner="DILU", tolerance=1e-30,
                  maxIter=iters - 1)
x, perf = m.solve(np.zeros(p["nCells"]), src[:, 0].copy(), solver="PBiCG", preconditioner="DILU", tolerance=1e-30,
                  maxIter=iters - 1)
print("scalar PBiCG: %.3f ms/it" % (1e3 * perf["solveSeconds"] / perf["nIterations"]))
