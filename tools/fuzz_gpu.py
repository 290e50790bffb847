"""Randomised stress of the bit-exact paths against the oracle (not a test: a hunt for rare failures).
python tools/fuzz_gpu.py [seconds] [seed]"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import conftest  # noqa
import oracle_py as O
from openfoam_amd import capi, cases

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
ctx = capi.Context(0)
t0 = time.time()
n_cases = n_checks = 0
while time.time() - t0 < budget:
    kind = rng.randint(6)   # (round 3: + wide-row graphs and small octree meshes - cooperative rows, lag buckets)
    asym = bool(rng.randint(2))
    if kind == 0:
        n = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 130, 257, 700, 2999, 3001, 5000]))
        p = cases.random_graph(n, int(rng.randint(1, 8)), max(1, min(n - 1, int(rng.randint(1, 400)))), asym=asym) if n > 1 \
            else dict(cases.box3d(1, 1, 1))
    elif kind == 1:
        p = cases.box3d(int(rng.randint(1, 40)), int(rng.randint(1, 40)), int(rng.randint(1, 40)), asym=asym)
    elif kind == 2:
        n = int(rng.randint(1000, 70000))
        p = cases.random_graph(n, int(rng.randint(2, 7)), int(rng.randint(5, 600)), asym=asym)
    elif kind == 3:
        n = int(rng.randint(200, 20000))
        p = cases.random_graph(n, int(rng.randint(6, 15)), int(rng.randint(20, n - 1)), asym=asym)
    elif kind == 4:
        n = int(rng.randint(3000, 40000))
        p = cases.random_graph(n, int(rng.randint(16, 44)), int(rng.randint(100, 900)), asym=asym)
    else:
        from openfoam_amd import octree
        b = int(rng.randint(1, 4))
        p = octree.problem(base=(5 * b, 2 * b, 2 * b), surface_levels=(int(rng.randint(3, 5)), 5), box_level=int(rng.randint(2, 4)))
        p.pop("cellLevel")
        if rng.randint(2):
            order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
            nl_, nu_, fm_, fl_ = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
            p = cases.renumbered(p, order, fm_, fl_, nl_, nu_)
    n = p["nCells"]
    psi, src = rng.randn(n), rng.randn(n)
    start = int(os.environ.get("FUZZ_START", "0"))
    if n_cases < start:   # replay: same random stream, nothing built
        rng.randint(1, 6)
        if "lower" in p and p["lowerAddr"].size:
            rng.randn(n, 3); rng.randn(n, 3)
        n_cases += 1
        continue
    S = O.System(p)
    a, m = capi.from_problem(ctx, p)
    checks = [("Amul", lambda: m.Amul(psi), lambda: S.Amul(psi)),
              ("residual", lambda: m.residual(psi, src), lambda: S.residual(psi, src))]
    k = int(rng.randint(1, 6))
    checks.append(("GS%d" % k, lambda: m.smooth("GaussSeidel", psi, src, k), lambda: S.smooth("GaussSeidel", psi, src, k)))
    checks.append(("symGS", lambda: m.smooth("symGaussSeidel", psi, src, 2), lambda: S.smooth("symGaussSeidel", psi, src, 2)))
    if S.sym:
        checks.append(("DIC", lambda: m.precondition("DIC", src), lambda: S.precondition("DIC", src)[0]))
    else:
        checks.append(("DILU", lambda: m.precondition("DILU", src), lambda: S.precondition("DILU", src)[0]))
        checks.append(("DILUT", lambda: m.precondition("DILU", src, transpose=True), lambda: S.precondition("DILU", src, transpose=True)[0]))
        if p["lowerAddr"].size:
            P3, S3 = rng.randn(n, 3), rng.randn(n, 3)
            checks.append(("cDILU", lambda: m.coupled_precondition("DILU", S3), lambda: S.c_precondition("DILU", S3)))
            checks.append(("cGS", lambda: m.coupled_smooth(P3, S3, 2), lambda: S.c_smooth(P3, S3, 2)))
    only = os.environ.get("FUZZ_ONLY")
    for name, g, o in checks:
        if only and name != only: continue
        for rep in range(2):
            try:
                got = g()
            except Exception as e:
                print("ERROR", name, "n", n, "faces", p["lowerAddr"].size, "asym", asym, "kind", kind, "rep", rep, "case", n_cases,
                      "info", a.info(), "->", e, flush=True)
                sys.exit(2)
            if not np.array_equal(got, o()):
                print("MISMATCH", name, "n", n, "faces", p["lowerAddr"].size, "asym", asym, "kind", kind, "rep", rep, flush=True)
                sys.exit(1)
            n_checks += 1
    m.close(); a.close()
    n_cases += 1
    if start and n_cases > start: break
print("fuzz ok: %d problems, %d bit-exact comparisons in %.0f s (seed %d)" % (n_cases, n_checks, time.time() - t0, seed))
