"""PCIe-inclusive cost of the drop-in boundary: the OpenFOAM shim hands over HOST arrays (pageable), every solve.
Times ldu_matrix_set_coeffs + ldu_solve with numpy arrays against device-resident inputs (bench.py's `value`)."""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import conftest  # noqa
from openfoam_amd import capi, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
p = cases.box3d(n)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10, mergeLevels=1,
          cacheAgglomeration=True, tolerance=1e-7, relTol=0.01)
psi0 = np.zeros(p["nCells"])
for rep in range(4):
    t0 = time.perf_counter()
    m.set_coeffs(p["diag"], p["upper"])
    ctx.sync()
    t1 = time.perf_counter()
    x, perf = m.solve(psi0, p["source"], history=False, **kw)
    t2 = time.perf_counter()
    print("rep %d: set_coeffs (host arrays, %.2f GB) %.1f ms; solve incl. psi/source up + psi down %.1f ms, of which on-device %.1f ms, %d V-cycles"
          % (rep, (p["diag"].nbytes + p["upper"].nbytes) / 1e9, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * perf["solveSeconds"], perf["nIterations"]), flush=True)
