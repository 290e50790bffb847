"""GPU (-m gpu): degenerate and boundary inputs through the C ABI, against the oracle (itself checked
against the reference on the same inputs in test_edge_cases_oracle_vs_reference): one cell, no faces,
zero source, exact initial guess, maxIter 0, a GAMG matrix too small to coarsen, star graphs with > 255
neighbours, malformed addressing."""
import numpy as np
import pytest

from openfoam_amd import capi

pytestmark = pytest.mark.gpu


def prob(nC, l, u, diag, upper, source, psi=None, lower=None):
    p = dict(nCells=nC, lowerAddr=np.array(l, dtype=np.int32), upperAddr=np.array(u, dtype=np.int32),
             diag=np.array(diag, float), upper=np.array(upper, float), source=np.array(source, float),
             psi=np.zeros(nC) if psi is None else np.array(psi, float))
    if lower is not None:
        p["lower"] = np.array(lower, float)
    return p


CASES = {
    "one_cell": prob(1, [], [], [2.0], [], [3.0]),
    "two_cells": prob(2, [0], [1], [2.0, 3.0], [-1.0], [1.0, 2.0]),
    "no_faces": prob(70, [], [], 1.0 + np.arange(70.0), [], np.sin(np.arange(70.0))),
    "zero_source": prob(3, [0, 1], [1, 2], [2, 2, 2], [-1, -1], [0, 0, 0]),
    "solved_already": prob(2, [0], [1], [2.0, 3.0], [-1.0], [0.0, 4.0], psi=[1.0, 2.0]),
    "asym_two": prob(2, [0], [1], [2.0, 3.0], [-1.0], [1.0, 2.0], lower=[-0.5]),
}
SOLVERS = [dict(solver="PCG", preconditioner="DIC"), dict(solver="PCG", preconditioner="diagonal"),
           dict(solver="PBiCG", preconditioner="DILU"), dict(solver="smoothSolver", smoother="GaussSeidel", maxIter=60),
           dict(solver="smoothSolver", smoother="symGaussSeidel", maxIter=60), dict(solver="diagonal")]


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_degenerate_systems(ctx, oracle, name):
    p = CASES[name]
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    assert np.array_equal(m.Amul(p["source"]), S.Amul(p["source"]))
    assert np.array_equal(m.residual(p["psi"], p["source"]), S.residual(p["psi"], p["source"]))
    for kw in SOLVERS:
        if kw["solver"] == "PCG" and "lower" in p:
            continue
        if kw["solver"] == "diagonal" and p["lowerAddr"].size:
            continue
        okw = dict(kw, tolerance=1e-10, relTol=0)
        gkw = dict(okw)
        if "preconditioner" in okw:
            okw["precond"] = okw.pop("preconditioner")
        x, perf = m.solve(p["psi"], p["source"], **gkw)
        xo, po = S.solve(p["psi"], p["source"], **okw)
        assert perf["nIterations"] == po["nIterations"], (name, kw)
        assert perf["converged"] == po["converged"] and perf["singular"] == po["singular"], (name, kw)
        np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]],
                                   [po["initialResidual"], po["finalResidual"]], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(x, xo, rtol=1e-12, atol=1e-300)
    m.close(); a.close()


def test_max_iter_zero_and_one(ctx, oracle):
    p = CASES["two_cells"]
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    for mi in (0, 1):
        x, perf = m.solve(p["psi"], p["source"], solver="smoothSolver", smoother="GaussSeidel", tolerance=1e-12,
                          relTol=0, maxIter=mi)
        xo, po = S.solve(p["psi"], p["source"], solver="smoothSolver", smoother="GaussSeidel", tolerance=1e-12,
                         relTol=0, maxIter=mi)
        assert perf["nIterations"] == po["nIterations"] and np.array_equal(x, xo)
    m.close(); a.close()


def test_gamg_on_a_matrix_too_small_to_coarsen_fails_loudly(ctx):
    """GAMGSolver.C:86-101 "No coarse levels created": the reference aborts, the library returns an error"""
    p = CASES["two_cells"]
    a, m = capi.from_problem(ctx, p)
    with pytest.raises(capi.LduError):
        m.solve(p["psi"], p["source"], solver="GAMG", smoother="GaussSeidel", agglomerator="algebraicPair",
                tolerance=1e-10, relTol=0)
    m.close(); a.close()


def test_malformed_addressing_is_rejected(ctx):
    with pytest.raises(capi.LduError):      # lower >= upper
        capi.Addressing(ctx, 3, np.array([1], dtype=np.int32), np.array([1], dtype=np.int32))
    with pytest.raises(capi.LduError):      # not sorted by owner
        capi.Addressing(ctx, 3, np.array([1, 0], dtype=np.int32), np.array([2, 1], dtype=np.int32))
    with pytest.raises(capi.LduError):      # neighbour out of range
        capi.Addressing(ctx, 3, np.array([0], dtype=np.int32), np.array([3], dtype=np.int32))
    nC = 300                                  # star: cell 0 owns 299 faces (> 255 entries per row)
    with pytest.raises(capi.LduError):
        capi.Addressing(ctx, nC, np.zeros(nC - 1, dtype=np.int32), np.arange(1, nC, dtype=np.int32))


def test_star_graph_at_the_row_width_limit(ctx, oracle):
    """255 upper neighbours of one cell / 255 lower neighbours of the last: the widest rows the layout holds"""
    n = 256
    l = np.concatenate([np.zeros(n - 1, dtype=np.int32), np.arange(1, n - 1, dtype=np.int32)])
    u = np.concatenate([np.arange(1, n, dtype=np.int32), np.full(n - 2, n - 1, dtype=np.int32)])
    order = np.lexsort((u, l))
    l, u = l[order], u[order]
    keep = np.ones(l.size, dtype=bool)
    keep[1:] = (l[1:] != l[:-1]) | (u[1:] != u[:-1])
    l, u = l[keep], u[keep]
    rng = np.random.RandomState(3)
    upper = -rng.rand(l.size)
    diag = np.zeros(n)
    np.subtract.at(diag, l, upper); np.subtract.at(diag, u, upper)
    diag += 1.0
    p = dict(nCells=n, lowerAddr=l, upperAddr=u, diag=diag, upper=upper, source=rng.randn(n), psi=np.zeros(n))
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    assert np.array_equal(m.Amul(p["source"]), S.Amul(p["source"]))
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], p["source"], 2), S.smooth("GaussSeidel", p["psi"], p["source"], 2))
    assert np.array_equal(m.precondition("DIC", p["source"]), S.precondition("DIC", p["source"])[0])
    m.close(); a.close()
