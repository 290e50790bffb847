"""which sweep aborts on the band-limited random graph (bench.py --mesh random at reduced size)?"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = cases.random_graph_fast(n, 7.0, 600)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
print("info", a.info(), "engines", [a.sweep_engine(k) for k in (0, 1, 2)], flush=True)
rng = np.random.RandomState(1)
x, b = rng.randn(n), rng.randn(n)
for name, fn in (("GS1", lambda: m.smooth("GaussSeidel", x, b, 1)), ("GS2", lambda: m.smooth("GaussSeidel", x, b, 2)),
                 ("DIC", lambda: m.precondition("DIC", b))):
    t0 = time.perf_counter(); f0 = ctx.fallback_count()
    fn()
    print(name, "%.1f ms" % (1e3 * (time.perf_counter() - t0)), "fallbacks", ctx.fallback_count() - f0, flush=True)
kw = dict(solver="GAMG", tolerance=1e-7, relTol=0.01, smoother="GaussSeidel", nPreSweeps=0, nPostSweeps=2, nFinestSweeps=2,
          cacheAgglomeration=1, agglomerator="faceAreaPair", nCellsInCoarsestLevel=10, mergeLevels=1)
t0 = time.perf_counter(); f0 = ctx.fallback_count()
xs, perf = m.solve(p["psi"], p["source"], **kw)
print("GAMG first solve %.2f s, %d V-cycles, fallbacks %d" % (time.perf_counter() - t0, perf["nIterations"], ctx.fallback_count() - f0), flush=True)
for L in m.gamg_level_sizes(**kw):
    print(L)
t0 = time.perf_counter(); f0 = ctx.fallback_count()
xs, perf = m.solve(p["psi"], p["source"], **kw)
print("GAMG second solve %.3f s, %d V-cycles, fallbacks %d" % (time.perf_counter() - t0, perf["nIterations"], ctx.fallback_count() - f0), flush=True)
