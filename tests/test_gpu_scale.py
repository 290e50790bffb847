"""GPU (-m gpu): medium-size cases (0.5-1 M cells) where the multi-workgroup paths, large levels and
deep GAMG hierarchies are exercised; still checked against the CPU oracle (seconds)."""
import numpy as np
import pytest

from openfoam_amd import capi, cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _cmp(x, perf, xo, po, long_krylov=False, bicg=False):
    """Short runs: the standard bar (equal count, history 1e-6 + 1e-12).  Long Krylov runs (hundreds
    of iterations): the 1e-16 differences of the tree-summed dot products are amplified by the
    recurrence, so only the first 50 iterations are held to 1e-6; afterwards the curves must stay
    within 15 % and the iteration counts within 2 % (stated tolerance)."""
    if not long_krylov:
        assert perf["nIterations"] == po["nIterations"]
        np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-6, atol=1e-12)
        assert np.max(np.abs(x - xo)) <= 1e-8 * np.max(np.abs(xo))
        return
    if bicg:
        # BiCG residuals are erratic (spikes of orders of magnitude): after the rounding differences
        # have been amplified the two curves are different realisations of the same irregular
        # convergence.  Held to: first 30 iterations at 1e-6, both converge, counts within 10 %.
        np.testing.assert_allclose(perf["history"][:30], po["history"][:30], rtol=1e-6, atol=1e-12)
        assert perf["converged"] and po["converged"]
        assert abs(perf["nIterations"] - po["nIterations"]) <= max(2, 0.10 * po["nIterations"])
        assert np.max(np.abs(x - xo)) <= 1e-5 * np.max(np.abs(xo))
        return
    assert abs(perf["nIterations"] - po["nIterations"]) <= max(1, 0.02 * po["nIterations"])
    n = min(len(perf["history"]), len(po["history"]))
    np.testing.assert_allclose(perf["history"][:50], po["history"][:50], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(perf["history"][:n], po["history"][:n], rtol=0.15, atol=1e-12)
    assert np.max(np.abs(x - xo)) <= 1e-5 * np.max(np.abs(xo))


def test_box_sym_gamg_and_pcg(ctx, oracle):
    p = cases.box3d(80)    # 512k cells, 238 dependency levels
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    src = p["source"]
    assert np.array_equal(m.Amul(src), S.Amul(src))
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], src, 2), S.smooth("GaussSeidel", p["psi"], src, 2))
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], src, 5), S.smooth("GaussSeidel", p["psi"], src, 5))
    assert np.array_equal(m.precondition("DIC", src), S.precondition("DIC", src)[0])
    kw = dict(solver="GAMG", smoother="GaussSeidel", tolerance=1e-8, relTol=0, cacheAgglomeration=1)
    okw = {k: v for k, v in kw.items() if k != "cacheAgglomeration"}
    _cmp(*m.solve(p["psi"], src, **kw), *S.solve(p["psi"], src, **okw))
    kw = dict(solver="PCG", preconditioner="DIC", tolerance=1e-7, relTol=0)
    _cmp(*m.solve(p["psi"], src, **kw), *S.solve(p["psi"], src, solver="PCG", precond="DIC", tolerance=1e-7, relTol=0),
         long_krylov=True)
    m.close(); a.close()


def test_box_asym_pbicg_gamg(ctx, oracle):
    p = cases.box3d(64, asym=True)
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    src = p["source"]
    assert np.array_equal(m.Tmul(src), S.Tmul(src))
    assert np.array_equal(m.precondition("DILU", src, transpose=True), S.precondition("DILU", src, transpose=True)[0])
    kw = dict(solver="PBiCG", preconditioner="DILU", tolerance=1e-8, relTol=0)
    _cmp(*m.solve(p["psi"], src, **kw), *S.solve(p["psi"], src, solver="PBiCG", precond="DILU", tolerance=1e-8, relTol=0),
         long_krylov=True, bicg=True)
    kw = dict(solver="GAMG", smoother="GaussSeidel", tolerance=1e-8, relTol=0)
    _cmp(*m.solve(p["psi"], src, **kw), *S.solve(p["psi"], src, **kw))
    kw = dict(solver="smoothSolver", smoother="GaussSeidel", nSweeps=3, tolerance=1e-3, relTol=0, maxIter=60)
    _cmp(*m.solve(p["psi"], src, **kw), *S.solve(p["psi"], src, **kw))
    m.close(); a.close()


def test_unstructured_wide_rows(ctx, oracle):
    """irregular graph, rows with up to ~20 entries: generic (non fast-path) entry loops, ragged slices"""
    p = cases.random_graph(200000, avg_deg=9, band=3000, seed=3)
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    src = p["source"]
    assert np.array_equal(m.Amul(src), S.Amul(src))
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], src, 3), S.smooth("GaussSeidel", p["psi"], src, 3))
    assert np.array_equal(m.smooth("symGaussSeidel", p["psi"], src, 1), S.smooth("symGaussSeidel", p["psi"], src, 1))
    assert np.array_equal(m.precondition("DIC", src), S.precondition("DIC", src)[0])
    kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="algebraicPair", tolerance=1e-8, relTol=0)
    _cmp(*m.solve(p["psi"], src, **kw), *S.solve(p["psi"], src, **kw))
    m.close(); a.close()


def test_2d_deep_narrow_dag(ctx, oracle):
    """damBreak-like 2-D case (C5 twin): 1198 dependency levels of <= 600 rows"""
    p = cases.jump2d(600, 600)
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    src = p["source"]
    assert np.array_equal(m.precondition("DIC", src), S.precondition("DIC", src)[0])
    kw = dict(solver="PCG", preconditioner="DIC", tolerance=1e-7, relTol=0.05)   # damBreak p_rgh settings
    x, perf = m.solve(p["psi"], src, **kw)
    xo, po = S.solve(p["psi"], src, solver="PCG", precond="DIC", tolerance=1e-7, relTol=0.05)
    _cmp(x, perf, xo, po, long_krylov=True)
    m.close(); a.close()
