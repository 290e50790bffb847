"""CPU: the coupled-solver oracle (oracle/ldu_oracle_coupled.c, LduMatrix<Type,scalar,scalar>) against the
golden vectors written by the REAL reference (tests/golden/coupled_*.npz) and, when oracle/_ref is built,
against the reference itself on fresh seeded problems - everything bit for bit, whole solves included
(every reduction is restated in the reference's left-to-right order)."""
import os

import numpy as np
import pytest

from openfoam_amd import cases

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(f for f in os.listdir(os.path.join(HERE, "golden")) if f.startswith("coupled_"))


def _check(S, n, psi, src, ref, sols, tol, maxIter):
    eq = lambda a, b: np.array_equal(np.asarray(a).ravel(), np.asarray(b).ravel())
    assert eq(S.c_ATmul(psi), ref["Amul"])
    assert eq(S.c_ATmul(psi, True), ref["Tmul"])
    assert eq(S.c_residual(psi, src), ref["residual"])
    for kind in ("DILU", "diagonal", "none"):
        if "precond_" + kind in ref:
            assert eq(S.c_precondition(kind, src), ref["precond_" + kind]), kind
        if "precondT_" + kind in ref:
            assert eq(S.c_precondition(kind, src, True), ref["precondT_" + kind]), kind
    x1 = S.c_smooth(psi, src, 1)
    assert eq(x1, ref["smooth1_GaussSeidel"])
    assert eq(S.c_smooth(x1, src, 2), ref["smooth3_GaussSeidel"])
    for (solver, pre), (xr, perf) in sols.items():
        x, po = S.c_solve(psi, src, solver=solver, preconditioner=pre, tolerance=tol, relTol=0.0,
                          maxIter=maxIter, nSweeps=2)
        assert eq(x, xr), (solver, pre)
        assert po["nIterations"] == int(perf[6]) and po["converged"] == bool(perf[7]), (solver, pre)
        assert np.array_equal(po["initialResidual"], perf[0:3]) and np.array_equal(po["finalResidual"], perf[3:6])


@pytest.mark.parametrize("fn", GOLDEN)
def test_coupled_oracle_vs_golden(fn, oracle):
    g = np.load(os.path.join(HERE, "golden", fn), allow_pickle=False)
    p = {k[2:]: g[k] for k in g.files if k.startswith("p_")}
    p["nCells"] = int(p["nCells"])
    n = p["nCells"]
    S = oracle.System(p)
    sols = {}
    for key in [k for k in g.files if k.startswith("solve_") and k.endswith("_psi")]:
        _, solver, pre, _ = key.split("_")
        sols[(solver, pre)] = (g[key], g["solve_%s_%s_perf" % (solver, pre)])
    assert len(sols) >= 3
    _check(S, n, g["psiV"].reshape(n, 3), g["sourceV"].reshape(n, 3), {k: g[k] for k in g.files}, sols,
           g["tolerance"], int(g["maxIter"]))


REF_PROBLEMS = {
    "box_asym_10": lambda: cases.box3d(10, asym=True),
    "rand_asym_500": lambda: cases.random_graph(500, asym=True),
    "rand_dense_asym_300": lambda: cases.random_graph(300, avg_deg=14, band=299, asym=True),
    "lap2d_30": lambda: cases.laplacian2d(30, 30),
    "jump2d_24": lambda: cases.jump2d(24, 24),
}


@pytest.mark.parametrize("name", sorted(REF_PROBLEMS))
def test_coupled_oracle_vs_reference(name, oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    p = REF_PROBLEMS[name]()
    rng = np.random.RandomState(4)
    n = p["nCells"]
    p["psiV"], p["sourceV"] = rng.randn(n * 3), rng.randn(n * 3)
    ref, _ = oracle.run_ref("cops", p)
    asym = "lower" in p
    combos = ([("PBiCCCG", "DILU"), ("PBiCICG", "DILU"), ("PBiCICG", "diagonal"), ("PBiCCCG", "none"),
               ("SmoothSolver", "none")] if asym else [("PCICG", "diagonal"), ("PCICG", "none"), ("SmoothSolver", "none")])
    tol = [1e-7, 1e-8, 1e-6]
    sols = {}
    for solver, pre in combos:
        d = ("solver %s; preconditioner %s; smoother GaussSeidel; tolerance (1e-7 1e-8 1e-6); relTol (0 0 0); "
             "maxIter 50; nSweeps 2;" % (solver, pre))
        r, _ = oracle.run_ref("csolve", p, d)
        sols[(solver, pre)] = (r["psiV"], r["perf"])
    _check(oracle.System(p), n, p["psiV"].reshape(n, 3), p["sourceV"].reshape(n, 3), ref, sols, tol, 50)


def test_coupled_selection_tables(oracle):
    """Names outside the matrix's table are fatal in the reference (LduMatrixSolver.C:58-70, :82-94;
    lduSolvers.C:33-50; lduPreconditioners.C:41-42): the oracle refuses them too."""
    sym, asym = cases.box3d(4), cases.box3d(4, asym=True)
    rng = np.random.RandomState(0)
    for p, bad in ((sym, [("PBiCCCG", "none"), ("PBiCICG", "none"), ("PCICG", "DILU")]), (asym, [("PCICG", "none")])):
        S = oracle.System(p)
        x, b = rng.randn(p["nCells"], 3), rng.randn(p["nCells"], 3)
        for solver, pre in bad:
            with pytest.raises(ValueError):
                S.c_solve(x, b, solver=solver, preconditioner=pre)
            if oracle.ref_available():
                q = dict(p, psiV=x.ravel(), sourceV=b.ravel())
                with pytest.raises(RuntimeError):
                    oracle.run_ref("csolve", q, "solver %s; preconditioner %s; tolerance (1e-6 1e-6 1e-6); "
                                   "relTol (0 0 0);" % (solver, pre))


def test_coupled_oracle_symmtensor_vs_reference(oracle):
    """Six components: LduMatrix<symmTensor, scalar, scalar>.  PBiCCCG's scalar products are the symmTensor
    double inner product (off-diagonal components count twice, SymmTensorI.H:212-220) - bit for bit."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    p = cases.box3d(7, 6, 5, asym=True)
    n = p["nCells"]
    rng = np.random.RandomState(8)
    p["psiV"], p["sourceV"] = rng.randn(n * 6), rng.randn(n * 6)
    tol = [1e-8, 1e-8, 1e-7, 1e-8, 1e-9, 1e-8]
    for solver in ("PBiCCCG", "PBiCICG", "SmoothSolver"):
        r, _ = oracle.run_ref("csolve6", p, "solver %s; preconditioner DILU; smoother GaussSeidel; nSweeps 2; "
                              "tolerance (1e-8 1e-8 1e-7 1e-8 1e-9 1e-8); relTol (0 0 0 0 0 0); maxIter 50;" % solver)
        x, perf = oracle.System(p).c_solve(p["psiV"].reshape(n, 6), p["sourceV"].reshape(n, 6), solver=solver,
                                           preconditioner="DILU", tolerance=tol, maxIter=50, nSweeps=2)
        assert np.array_equal(x.ravel(), r["psiV"]), solver
        assert perf["nIterations"] == int(r["perf"][12]) and np.array_equal(perf["finalResidual"], r["perf"][6:12])
