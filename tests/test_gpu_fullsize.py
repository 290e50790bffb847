"""GPU (-m gpu): BASELINE.json's full-size workload (216^3 box, 10 077 696 cells, 30 093 120 faces) against
the CPU oracle: kernels bit for bit, the benchmark's GAMG p-solve and 25 PCG/DIC iterations by history.
About 40 s (the oracle needs ~1 s per sweep / V-cycle at this size)."""
import numpy as np
import pytest

from openfoam_amd import capi, cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    p = cases.box3d(216)
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, p)
    yield p, a, m
    m.close(); a.close(); ctx.close()


def test_engines_at_full_size(full):
    p, a, m = full
    assert a.sweep_engine(0) == "clusters" and a.sweep_engine(2) == "clusters"


def test_kernels_bitexact_at_full_size(full, oracle):
    p, a, m = full
    S = oracle.System(p)
    src = p["source"]
    assert np.array_equal(m.Amul(src), S.Amul(src))
    assert np.array_equal(m.residual(src, src), S.residual(src, src))
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], src, 1), S.smooth("GaussSeidel", p["psi"], src, 1))
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], src, 2), S.smooth("GaussSeidel", p["psi"], src, 2))
    assert np.array_equal(m.precondition("DIC", src), S.precondition("DIC", src)[0])


def test_benchmark_solve_history_at_full_size(full, oracle):
    p, a, m = full
    S = oracle.System(p)
    kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
              mergeLevels=1, tolerance=1e-7, relTol=0.01)
    x, perf = m.solve(p["psi"], p["source"], cacheAgglomeration=1, **kw)
    xo, po = S.solve(p["psi"], p["source"], **kw)
    assert perf["nIterations"] == po["nIterations"] == 4
    np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-6, atol=1e-12)
    assert np.max(np.abs(x - xo)) <= 1e-8 * np.max(np.abs(xo))
    x, perf = m.solve(p["psi"], p["source"], solver="PCG", preconditioner="DIC", tolerance=0, relTol=0, maxIter=25)
    xo, po = S.solve(p["psi"], p["source"], solver="PCG", precond="DIC", tolerance=0, relTol=0, maxIter=25)
    np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-6, atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config C5 at its stated size: interFoam damBreak refined to ~4 M cells (blockMeshDict:49-53 x 42^2), 2-D,
# two-phase coefficient jump 1000.  The worst case for every sweep engine: ~4000 dependency levels of <= 2000 rows.
@pytest.fixture(scope="module")
def jump():
    p = cases.jump2d(2000, 2000)
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, p)
    yield p, a, m, ctx
    m.close(); a.close(); ctx.close()


def test_c5_kernels_bitexact_at_4M_cells(jump, oracle):
    p, a, m, ctx = jump
    assert p["nCells"] == 4000000 and a.info()["nLevels"] == 3999
    S = oracle.System(p)
    src = p["source"]
    assert np.array_equal(m.Amul(src), S.Amul(src))
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], src, 1), S.smooth("GaussSeidel", p["psi"], src, 1))
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], src, 4), S.smooth("GaussSeidel", p["psi"], src, 4))
    assert np.array_equal(m.precondition("DIC", src), S.precondition("DIC", src)[0])
    assert ctx.fallback_count() == 0


def test_c5_solves_at_4M_cells(jump, oracle):
    """GAMG p_rgh (BASELINE's wording; the motorBike GAMG block) and the tutorial's own PCG/DIC (damBreak fvSolution:
    tolerance 1e-7, relTol 0.05) by residual history"""
    p, a, m, ctx = jump
    S = oracle.System(p)
    kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
              mergeLevels=1, tolerance=1e-7, relTol=0.01)
    x, perf = m.solve(p["psi"], p["source"], cacheAgglomeration=1, **kw)
    xo, po = S.solve(p["psi"], p["source"], **kw)
    assert perf["nIterations"] == po["nIterations"]
    np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-6, atol=1e-12)
    assert np.max(np.abs(x - xo)) <= 1e-8 * np.max(np.abs(xo))
    for kw in (dict(tolerance=1e-7, relTol=0.05, maxIter=1000), dict(tolerance=0, relTol=0, maxIter=25)):
        x, perf = m.solve(p["psi"], p["source"], solver="PCG", preconditioner="DIC", **kw)
        xo, po = S.solve(p["psi"], p["source"], solver="PCG", precond="DIC", **kw)
        assert perf["nIterations"] == po["nIterations"]
        np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-6, atol=1e-12)
    assert ctx.fallback_count() == 0
