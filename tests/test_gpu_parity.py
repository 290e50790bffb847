"""GPU parity (-m gpu): the HIP path through the C ABI (libldugpu.so) against the CPU oracle
on the same seeded inputs, and against the golden vectors produced by the real reference.

Bars:
  * matrix ops, preconditioner applications and smoother sweeps: BIT-EXACT (level scheduling
    keeps every row's accumulation order; no FMA contraction);
  * whole solves: identical iteration count; residual history (normalised residuals, initial
    residual O(1)) within |h_gpu - h_ref| <= 1e-6*h_ref + 1e-12.  Only the global sums differ
    (tree vs left-to-right accumulation, ~1e-16 relative per sum); Krylov recurrences amplify
    that as the residual drops, hence the absolute floor.  Un-preconditioned CG (no
    convergence, orthogonality lost) is compared at 1e-2.
"""
import os
import sys

import numpy as np
import pytest

from openfoam_amd import capi, cases, ldub

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
HIST_RTOL = 1e-6
HIST_ATOL = 1e-12

PROBLEMS = {
    "lap2d_40": lambda: cases.laplacian2d(40, 40),
    "box3d_12": lambda: cases.box3d(12),
    "box3d_20x7x5": lambda: cases.box3d(20, 7, 5),
    "box3d_asym_10": lambda: cases.box3d(10, asym=True),
    "rand_600": lambda: cases.random_graph(600),
    "rand_asym_500": lambda: cases.random_graph(500, asym=True),
    "rand_dense_300": lambda: cases.random_graph(300, avg_deg=14, band=299),
    "jump2d_24": lambda: cases.jump2d(24, 24),
    "tiny_3": lambda: cases.box3d(3, 1, 1),
    "diag_only": lambda: dict(cases.box3d(4, 1, 1), lowerAddr=np.zeros(0, np.int32),
                              upperAddr=np.zeros(0, np.int32), upper=np.zeros(0),
                              faceWeights=np.zeros(0)),
}


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module", params=sorted(PROBLEMS))
def prob(request):
    p = PROBLEMS[request.param]()
    rng = np.random.RandomState(1)
    p["psi"] = rng.randn(p["nCells"])
    p["source"] = rng.randn(p["nCells"])
    return p


def test_ops_bitexact(prob, ctx, oracle):
    S = oracle.System(prob)
    a, m = capi.from_problem(ctx, prob)
    psi, src = prob["psi"], prob["source"]
    assert np.array_equal(m.Amul(psi), S.Amul(psi))
    assert np.array_equal(m.Tmul(psi), S.Tmul(psi))
    assert np.array_equal(m.sumA(), S.sumA())
    assert np.array_equal(m.residual(psi, src), S.residual(psi, src))
    assert np.array_equal(m.H(psi), S.dom_op("orc_H", psi))
    assert np.array_equal(m.H1(), S.dom_op("orc_H1"))
    assert np.array_equal(m.faceH(psi), S.dom_op("orc_faceH", psi, out_faces=True))
    if S.sym:
        assert np.array_equal(m.precondition("DIC", src), S.precondition("DIC", src)[0])
        assert np.array_equal(m.precondition("FDIC", src), S.precondition("DIC", src)[0])
        smoothers = ["GaussSeidel", "symGaussSeidel", "DIC", "FDIC", "DICGaussSeidel"]
    else:
        assert np.array_equal(m.precondition("DILU", src), S.precondition("DILU", src)[0])
        assert np.array_equal(m.precondition("DILU", src, transpose=True),
                              S.precondition("DILU", src, transpose=True)[0])
        smoothers = ["GaussSeidel", "symGaussSeidel", "DILU", "DILUGaussSeidel"]
    assert np.array_equal(m.precondition("diagonal", src), src * (1.0 / prob["diag"]))
    for sm in smoothers:
        for n in (1, 3):
            assert np.array_equal(m.smooth(sm, psi, src, n), S.smooth(sm, psi, src, n)), (sm, n)
    # reductions: tree vs serial order -> tolerance
    assert abs(m.gSumProd(psi, src) - S.gSumProd(psi, src)) <= 1e-12 * np.sum(np.abs(psi * src))
    assert abs(m.gSumMag(psi) - S.gSumMag(psi)) <= 1e-12 * np.sum(np.abs(psi))
    m.close(); a.close()


SOLVES = [
    (dict(solver="PCG", preconditioner="DIC", tolerance=1e-10, relTol=0), "sym"),
    (dict(solver="PCG", preconditioner="FDIC", tolerance=1e-9, relTol=0), "sym"),
    (dict(solver="PCG", preconditioner="diagonal", tolerance=1e-8, relTol=0), "sym"),
    (dict(solver="PCG", preconditioner="none", tolerance=1e-6, relTol=0, maxIter=40), "sym"),
    (dict(solver="PBiCG", preconditioner="DILU", tolerance=1e-10, relTol=0), "asym"),
    (dict(solver="PBiCG", preconditioner="diagonal", tolerance=1e-8, relTol=0), "asym"),
    (dict(solver="smoothSolver", smoother="GaussSeidel", nSweeps=2, tolerance=1e-6, relTol=0, maxIter=100), "any"),
    (dict(solver="smoothSolver", smoother="symGaussSeidel", nSweeps=1, tolerance=1e-6, relTol=0, maxIter=60), "any"),
    (dict(solver="smoothSolver", smoother="GaussSeidel", nSweeps=-3), "any"),
    (dict(solver="diagonal"), "any"),
]


def _compare_solve(x, perf, xo, po, rtol=HIST_RTOL, xtol=1e-8):
    assert perf["nIterations"] == po["nIterations"]
    assert perf["converged"] == po["converged"]
    n = len(po["history"])
    assert len(perf["history"]) == n
    np.testing.assert_allclose(perf["history"], po["history"], rtol=rtol, atol=HIST_ATOL)
    scale = np.max(np.abs(xo)) + 1e-300
    assert np.max(np.abs(x - xo)) <= xtol * scale


@pytest.mark.parametrize("case", SOLVES, ids=["%s_%d" % (c[0]["solver"], i) for i, c in enumerate(SOLVES)])
def test_solve_history(prob, ctx, oracle, case):
    kw, kind = case
    S = oracle.System(prob)
    if kind == "sym" and not S.sym or kind == "asym" and S.sym:
        pytest.skip("matrix symmetry")
    if prob["lowerAddr"].size == 0 and kw["solver"] not in ("diagonal",):
        pytest.skip("diagonal-only matrix")
    okw = dict(kw)
    if "preconditioner" in okw:
        okw["precond"] = okw.pop("preconditioner")
    psi0 = np.zeros(prob["nCells"])
    xo, po = S.solve(psi0, prob["source"], **okw)
    a, m = capi.from_problem(ctx, prob)
    x, perf = m.solve(psi0, prob["source"], **kw)
    if kw.get("preconditioner") == "none":
        _compare_solve(x, perf, xo, po, rtol=1e-2, xtol=1e-3)
    else:
        _compare_solve(x, perf, xo, po)
    m.close(); a.close()


GAMG = [
    dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
         mergeLevels=1, tolerance=1e-9, relTol=0),
    dict(solver="GAMG", smoother="GaussSeidel", agglomerator="algebraicPair", nCellsInCoarsestLevel=20,
         mergeLevels=2, tolerance=1e-8, relTol=0, nPreSweeps=1),
    dict(solver="GAMG", smoother="symGaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
         mergeLevels=1, tolerance=1e-8, relTol=0, interpolateCorrection=1, nFinestSweeps=1, nPostSweeps=1),
    dict(solver="GAMG", smoother="DICGaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
         mergeLevels=1, tolerance=1e-8, relTol=0, cacheAgglomeration=1),
    # directSolveCoarsest (GAMGSolver.C:95-106): the coarsest level by the device LU (ldu_coarsest.hip)
    dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=30,
         mergeLevels=1, tolerance=1e-9, relTol=0, directSolveCoarsest=1),
]


@pytest.mark.parametrize("kw", GAMG, ids=["gamg%d" % i for i in range(len(GAMG))])
def test_gamg(prob, ctx, oracle, kw):
    if prob["nCells"] < 40:
        pytest.skip("too small for GAMG")
    S = oracle.System(prob)
    if not S.sym and "DIC" in kw["smoother"]:
        kw = dict(kw, smoother="DILUGaussSeidel")
    okw = {k: v for k, v in kw.items() if k != "cacheAgglomeration"}
    a, m = capi.from_problem(ctx, prob)
    lv = m.gamg_levels(**kw)
    lo = S.gamg_levels(**okw)
    assert [L["nCells"] for L in lv] == [L["nCells"] for L in lo]
    for Ld, Lo in zip(lv, lo):
        assert np.array_equal(Ld["restrict"], Lo["restrict"])
        assert np.array_equal(Ld["diag"], Lo["diag"])          # bit-exact coarse matrices
        assert np.array_equal(Ld["upper"], Lo["upper"])
        if not S.sym:
            assert np.array_equal(Ld["lower"], Lo["lower"])
    psi0 = np.zeros(prob["nCells"])
    xo, po = S.solve(psi0, prob["source"], **okw)
    x, perf = m.solve(psi0, prob["source"], **kw)
    _compare_solve(x, perf, xo, po)
    # second solve on the same matrix object (cached hierarchy / refreshed coefficients)
    m.set_coeffs(prob["diag"] * 1.5, prob["upper"], prob.get("lower"))
    p2 = dict(prob, diag=prob["diag"] * 1.5)
    xo2, po2 = oracle.System(p2).solve(psi0, prob["source"], **okw)
    x2, perf2 = m.solve(psi0, prob["source"], **kw)
    _compare_solve(x2, perf2, xo2, po2)
    m.close(); a.close()


def test_gamg_direct_solve_coarsest(ctx, oracle):
    """directSolveCoarsest on symmetric and asymmetric matrices, coarsest levels of 6..60 cells: histories and psi as the
    oracle's (whose LU is pinned bit-for-bit on the reference's, tests/test_oracle_vs_ref.py); one V-cycle of a
    2-level hierarchy is compared bit-for-bit through psi after maxIter=1."""
    for p, nC in ((cases.box3d(12), 10), (cases.box3d(12, asym=True), 40), (cases.irregular_box(9), 30),
                  (cases.box3d(16, asym=True), 64)):
        S = oracle.System(p)
        kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=nC, mergeLevels=1,
                  tolerance=1e-10, relTol=0, directSolveCoarsest=1)
        a, m = capi.from_problem(ctx, p)
        assert m.gamg_levels(**kw)[-1]["nCells"] <= 64
        psi0 = np.zeros(p["nCells"])
        xo, po = S.solve(psi0, p["source"], **kw)
        x, perf = m.solve(psi0, p["source"], **kw)
        _compare_solve(x, perf, xo, po)
        assert perf["nIterations"] == po["nIterations"]
        m.close(); a.close()
    # a coarsest level beyond the device LU's size is refused, loudly (no iterative stand-in)
    p = cases.box3d(12)
    a, m = capi.from_problem(ctx, p)
    with pytest.raises(capi.LduError, match="directSolveCoarsest"):
        m.solve(np.zeros(p["nCells"]), p["source"], solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                nCellsInCoarsestLevel=200, mergeLevels=1, directSolveCoarsest=1)
    m.close(); a.close()


def test_pcg_gamg_preconditioner(ctx, oracle):
    p = cases.box3d(10)
    S = oracle.System(p)
    kw = dict(solver="PCG", preconditioner="GAMG", smoother="GaussSeidel", tolerance=1e-9, relTol=0, nVcycles=2)
    okw = dict(kw, precond="GAMG"); okw.pop("preconditioner")
    xo, po = S.solve(p["psi"], p["source"], **okw)
    a, m = capi.from_problem(ctx, p)
    x, perf = m.solve(p["psi"], p["source"], **kw)
    _compare_solve(x, perf, xo, po)
    m.close(); a.close()


sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402


@pytest.mark.parametrize("name", sorted(make_golden.PROBLEMS))
def test_against_reference_golden(name, ctx):
    """HIP path vs the REAL reference's outputs (tests/golden, generated by make_golden.py)."""
    gen, solves = make_golden.PROBLEMS[name]
    p = gen()
    G = ldub.read(os.path.join(HERE, "golden", name + ".ldub"))
    a, m = capi.from_problem(ctx, p)
    assert np.array_equal(m.Amul(p["psi"]), G["ops_Amul"])
    assert np.array_equal(m.residual(p["psi"], p["source"]), G["ops_residual"])
    assert np.array_equal(m.smooth("GaussSeidel", p["psi"], p["source"], 1), G["ops_smooth1_GaussSeidel"])
    assert np.array_equal(m.smooth("symGaussSeidel", p["psi"], p["source"], 1), G["ops_smooth1_symGaussSeidel"])
    if "lower" not in p:
        assert np.array_equal(m.precondition("DIC", p["source"]), G["ops_precond_DIC"])
    else:
        assert np.array_equal(m.precondition("DILU", p["source"]), G["ops_precond_DILU"])
        assert np.array_equal(m.precondition("DILU", p["source"], transpose=True), G["ops_precondT_DILU"])
    for i, (sname, kw) in enumerate(solves):
        x, perf = m.solve(p["psi"], p["source"], **kw)
        gp = G["solve%d_perf" % i]
        assert perf["nIterations"] == int(gp[2]), (name, sname)
        np.testing.assert_allclose(perf["initialResidual"], gp[0], rtol=1e-10)
        np.testing.assert_allclose(perf["finalResidual"], gp[1], rtol=1e-6, atol=HIST_ATOL)
        h = G["solve%d_hist" % i]
        n = min(len(h), len(perf["history"]))
        np.testing.assert_allclose(perf["history"][:n], h[:n], rtol=1e-6, atol=HIST_ATOL)
        xo = G["solve%d_psi" % i]
        assert np.max(np.abs(x - xo)) <= 1e-7 * (np.max(np.abs(xo)) + 1e-300)
    m.close(); a.close()


def test_level_schedule_large_levels(ctx, oracle):
    """A case whose levels exceed the fuse threshold, so the per-level multi-block kernels run."""
    p = cases.box3d(40)   # 64k cells, plane levels up to ~1.2k rows
    os.environ["LDU_FUSE_ROWS"] = "256"
    c2 = capi.Context(0)
    try:
        S = oracle.System(p)
        a, m = capi.from_problem(c2, p)
        info = a.info()
        assert info["nLevels"] == 40 * 3 - 2
        assert np.array_equal(m.precondition("DIC", p["source"]), S.precondition("DIC", p["source"])[0])
        assert np.array_equal(m.smooth("GaussSeidel", p["psi"], p["source"], 2),
                              S.smooth("GaussSeidel", p["psi"], p["source"], 2))
        x, perf = m.solve(p["psi"], p["source"], solver="PCG", preconditioner="DIC", tolerance=1e-8, relTol=0)
        xo, po = S.solve(p["psi"], p["source"], solver="PCG", precond="DIC", tolerance=1e-8, relTol=0)
        _compare_solve(x, perf, xo, po)
        m.close(); a.close()
    finally:
        del os.environ["LDU_FUSE_ROWS"]
        c2.close()


@pytest.mark.parametrize("name", ["fvsolve2_halves_6x8x7", "fvsolve4_chain_5x6x6", "fvsolve3_chain_asym_5x7x6",
                                  "fvsolve3_chain_nonblocking_4x7x6", "fvsolve8_blocks_2x2x2_4x4x4",
                                  "fvsolve2_split_halves_5x6x6", "fvsolve4_blocks_2x2x1_split_4x4x5"])
def test_cyclic_patches_against_reference(ctx, name):
    """cyclic coupled patches on ONE rank (ldu_addr_add_cyclic_patch): the reference's own single-process
    solves with real cyclic patches (tests/golden/fvsolve*.npz), GAMG + Krylov."""
    from test_fv_oracle_golden import load, cyclic_problem, SMOOTHER
    g = load(name)
    sp = cyclic_problem(g)
    nB = int(g["nBoxes"]) if "nBoxes" in g else 2
    a = capi.Addressing(ctx, sp["nCells"], sp["lowerAddr"], sp["upperAddr"], sp["faceWeights"],
                        patches=sp["patches_dev"])
    m = capi.Matrix(a)
    m.set_coeffs(sp["diag"], sp["upper"], sp.get("lower"))
    for i, q in enumerate(sp["patches"]):
        m.set_patch_coeffs(i, q["bouCoeffs"], q["intCoeffs"])
    x, perf = m.solve(sp["psi"], sp["source"], solver="GAMG", smoother=SMOOTHER(name), agglomerator="faceAreaPair",
                      nCellsInCoarsestLevel=10 * nB, mergeLevels=1, tolerance=1e-10, relTol=0)
    r = g["ref_gamg_perf"]
    assert perf["nIterations"] == int(r[2]) and perf["converged"]
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-8 * np.max(np.abs(g["ref_gamg_psi"]))
    kw = dict(solver="PBiCG", preconditioner="DILU") if "asym" in name else dict(solver="PCG", preconditioner="DIC")
    x, perf = m.solve(sp["psi"], sp["source"], tolerance=1e-10, relTol=0, **kw)
    r = g["ref_pcg_perf"]
    assert perf["nIterations"] == int(r[2])
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_pcg_psi"])) <= 1e-8 * np.max(np.abs(g["ref_pcg_psi"]))
    m.close(); a.close()


@pytest.mark.parametrize("name", ["fvsolve4_chain_lu_5x6x6", "fvsolve3_chain_asym_lu_5x7x6", "fvsolve8_blocks_lu_2x2x2_4x4x4"])
def test_direct_solve_coarsest_with_cyclic_patches(ctx, oracle, name):
    """directSolveCoarsest with cyclic patches on ONE rank (round 6; LUscalarMatrix.C:128-187: the interfaces' coefficients
    subtracted from the dense matrix): the reference's own solve (golden) and the oracle's history.  40 / 30 cells: one row
    per lane; the 8-box fixture's coarsest level has more than 64 cells: two rows per lane (dense_lu_kernel<2>)."""
    from test_fv_oracle_golden import load, cyclic_problem
    g = load(name)
    sp = cyclic_problem(g)
    nB = int(g["nBoxes"]) if "nBoxes" in g else 2
    kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10 * nB, mergeLevels=1,
              tolerance=1e-10, relTol=0, directSolveCoarsest=1)
    a = capi.Addressing(ctx, sp["nCells"], sp["lowerAddr"], sp["upperAddr"], sp["faceWeights"], patches=sp["patches_dev"])
    m = capi.Matrix(a)
    m.set_coeffs(sp["diag"], sp["upper"], sp.get("lower"))
    for i, q in enumerate(sp["patches"]):
        m.set_patch_coeffs(i, q["bouCoeffs"], q["intCoeffs"])
    x, perf = m.solve(sp["psi"], sp["source"], history=True, **kw)
    xo, po = oracle.System([sp]).solve(sp["psi"], sp["source"], **kw)
    r = g["ref_gamg_perf"]
    assert perf["nIterations"] == int(r[2]) == po["nIterations"] and perf["converged"]
    np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose([perf["initialResidual"], perf["finalResidual"]], r[:2], rtol=1e-6)
    assert np.max(np.abs(x - g["ref_gamg_psi"])) <= 1e-8 * np.max(np.abs(g["ref_gamg_psi"]))
    # a second solve with new coefficients: the gathered matrix follows them
    m.set_coeffs(1.5 * sp["diag"], sp["upper"], sp.get("lower"))
    for i, q in enumerate(sp["patches"]):
        m.set_patch_coeffs(i, q["bouCoeffs"], q["intCoeffs"])
    sp2 = dict(sp, diag=1.5 * sp["diag"])
    x2, p2 = m.solve(sp["psi"], sp["source"], history=True, **kw)
    xo2, po2 = oracle.System([sp2]).solve(sp["psi"], sp["source"], **kw)
    assert p2["nIterations"] == po2["nIterations"]
    np.testing.assert_allclose(p2["history"], po2["history"], rtol=1e-6, atol=1e-12)
    assert np.max(np.abs(x2 - xo2)) <= 1e-8 * np.max(np.abs(xo2))
    m.close(); a.close()


@pytest.mark.parametrize("name", ["fvsolve2_halves_6x8x7", "fvsolve3_chain_asym_5x7x6", "fvsolve3_chain_nonblocking_4x7x6"])
def test_smoothers_with_cyclic_patches_bitexact(ctx, name):
    """GaussSeidel / nonBlockingGaussSeidel with coupled (cyclic) patches: bit-exact against the reference's
    own smoother classes (3 sweeps; the nonblocking fixture is the one where the two differ)."""
    from test_fv_oracle_golden import load, cyclic_problem
    g = load(name)
    sp = cyclic_problem(g)
    a = capi.Addressing(ctx, sp["nCells"], sp["lowerAddr"], sp["upperAddr"], sp["faceWeights"],
                        patches=sp["patches_dev"])
    m = capi.Matrix(a)
    m.set_coeffs(sp["diag"], sp["upper"], sp.get("lower"))
    for i, q in enumerate(sp["patches"]):
        m.set_patch_coeffs(i, q["bouCoeffs"], q["intCoeffs"])
    for sm in ("GaussSeidel", "nonBlockingGaussSeidel"):
        assert np.array_equal(m.smooth(sm, g["smooth_x0"], sp["source"], 3), g["ref_smooth_" + sm]), sm
    m.close(); a.close()
