"""Pin the C restatement (oracle/libldu_oracle.so) against the REAL reference
(oracle/_ref/libOpenFOAM.so built from /root/reference by oracle/build_ref.sh).

Kernel-level results must be bit-identical (same loop order, no FMA); whole
solves must reproduce the reference's residual history bit-for-bit as well,
since every operation is restated in the reference's order.
Skipped when the reference build is not present.
"""
import numpy as np
import pytest

from openfoam_amd import cases

import oracle_py

pytestmark = pytest.mark.skipif(not oracle_py.ref_available(),
                                reason="oracle/_ref not built (needs /root/reference)")

PROBLEMS = {
    "lap2d_40": lambda: cases.laplacian2d(40, 40),
    "box3d_12": lambda: cases.box3d(12),
    "box3d_asym_10": lambda: cases.box3d(10, asym=True),
    "rand_600": lambda: cases.random_graph(600),
    "rand_asym_500": lambda: cases.random_graph(500, asym=True),
    "jump2d_24": lambda: cases.jump2d(24, 24),
}


@pytest.fixture(scope="module", params=sorted(PROBLEMS))
def prob(request):
    p = PROBLEMS[request.param]()
    rng = np.random.RandomState(1)
    p["psi"] = rng.randn(p["nCells"])
    p["source"] = rng.randn(p["nCells"])
    return p


def test_ops_bitexact(prob, oracle):
    ref, _ = oracle.run_ref("ops", prob)
    S = oracle.System(prob)
    psi, src = prob["psi"], prob["source"]
    lo, os_, ls = S.addressing()
    assert np.array_equal(lo, ref["losort"])
    assert np.array_equal(os_, ref["ownerStart"])
    assert np.array_equal(ls, ref["losortStart"])
    assert np.array_equal(S.Amul(psi), ref["Amul"])
    assert np.array_equal(S.Tmul(psi), ref["Tmul"])
    assert np.array_equal(S.sumA(), ref["sumA"])
    assert np.array_equal(S.residual(psi, src), ref["residual"])
    assert np.array_equal(S.dom_op("orc_H", psi), ref["H"])
    assert np.array_equal(S.dom_op("orc_H1"), ref["H1"])
    assert np.array_equal(S.dom_op("orc_faceH", psi, out_faces=True), ref["faceH"])
    if S.sym:
        w, rD = S.precondition("DIC", src)
        assert np.array_equal(rD, ref["rD_DIC"])
        assert np.array_equal(w, ref["precond_DIC"])
        smoothers = ["GaussSeidel", "symGaussSeidel", "DIC", "FDIC", "DICGaussSeidel"]
    else:
        w, _ = S.precondition("DILU", src)
        assert np.array_equal(w, ref["precond_DILU"])
        wT, _ = S.precondition("DILU", src, transpose=True)
        assert np.array_equal(wT, ref["precondT_DILU"])
        smoothers = ["GaussSeidel", "symGaussSeidel", "DILU", "DILUGaussSeidel"]
    for sm in smoothers:
        x1 = S.smooth(sm, psi, src, 1)
        assert np.array_equal(x1, ref["smooth1_" + sm]), sm
        x3 = S.smooth(sm, x1, src, 2)
        assert np.array_equal(x3, ref["smooth3_" + sm]), sm


SOLVES = [
    ("PCG", dict(solver="PCG", preconditioner="DIC", tolerance=1e-10, relTol=0), "DICPCG", True),
    ("PCG", dict(solver="PCG", preconditioner="FDIC", tolerance=1e-9, relTol=0), "FDICPCG", True),
    ("PCG", dict(solver="PCG", preconditioner="diagonal", tolerance=1e-8, relTol=0), "diagonalPCG", True),
    ("PCG", dict(solver="PCG", preconditioner="none", tolerance=1e-6, relTol=0, maxIter=50), "nonePCG", True),
    ("PBiCG", dict(solver="PBiCG", preconditioner="DILU", tolerance=1e-10, relTol=0), "DILUPBiCG", False),
    ("PBiCG", dict(solver="PBiCG", preconditioner="diagonal", tolerance=1e-8, relTol=0), "diagonalPBiCG", False),
    ("smooth", dict(solver="smoothSolver", smoother="GaussSeidel", nSweeps=2, tolerance=1e-6,
                    relTol=0, maxIter=100), "smoothSolver", None),
    ("smooth", dict(solver="smoothSolver", smoother="symGaussSeidel", nSweeps=1, tolerance=1e-6,
                    relTol=0, maxIter=60), "smoothSolver", None),
]


@pytest.mark.parametrize("case", SOLVES, ids=[c[2] + "_" + str(i) for i, c in enumerate(SOLVES)])
def test_solve_history_bitexact(prob, oracle, case):
    _, kw, name, symOnly = case
    S = oracle.System(prob)
    if symOnly is True and not S.sym:
        pytest.skip("symmetric-only")
    if symOnly is False and S.sym:
        pytest.skip("asymmetric-only")
    p = dict(prob)
    p["psi"] = np.zeros(p["nCells"])
    ref, out = oracle.run_ref("solve", p, oracle.dict_string(**kw))
    x, perf = S.solve(p["psi"], p["source"], **kw)
    rp = ref["perf"]
    assert perf["nIterations"] == int(rp[2])
    assert perf["initialResidual"] == rp[0]
    assert perf["finalResidual"] == rp[1]
    assert perf["converged"] == bool(rp[3])
    assert np.array_equal(x, ref["psi"])
    hist = oracle.parse_history(out, name)
    # the reference prints one line per checkConvergence call
    n = min(len(hist), len(perf["history"]))
    assert n >= 1 and np.array_equal(hist[:n], perf["history"][:n])


GAMG = [
    dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
         mergeLevels=1, cacheAgglomeration=False, tolerance=1e-9, relTol=0),
    dict(solver="GAMG", smoother="GaussSeidel", agglomerator="algebraicPair", nCellsInCoarsestLevel=20,
         mergeLevels=2, cacheAgglomeration=False, tolerance=1e-8, relTol=0, nPreSweeps=1),
    dict(solver="GAMG", smoother="symGaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
         mergeLevels=1, cacheAgglomeration=False, tolerance=1e-8, relTol=0, interpolateCorrection=True,
         nFinestSweeps=1, nPostSweeps=1),
    dict(solver="GAMG", smoother="DICGaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
         mergeLevels=1, cacheAgglomeration=False, tolerance=1e-8, relTol=0),
]


def _okw(kw):
    kw = dict(kw)
    kw.pop("cacheAgglomeration", None)
    for k in ("interpolateCorrection",):
        if k in kw:
            kw[k] = int(kw[k])
    return kw


@pytest.mark.parametrize("kw", GAMG, ids=["gamg%d" % i for i in range(len(GAMG))])
def test_gamg_bitexact(prob, oracle, kw):
    S = oracle.System(prob)
    if not S.sym and "DIC" in kw["smoother"]:
        kw = dict(kw, smoother="DILUGaussSeidel")
    p = dict(prob)
    p["psi"] = np.zeros(p["nCells"])
    ds = oracle.dict_string(**kw)
    agg, _ = oracle.run_ref("agglom", p, ds)
    levels = S.gamg_levels(**_okw(kw))
    assert len(levels) == int(agg["nLevels"][0])
    for i, L in enumerate(levels):
        assert L["nCells"] == int(agg["nCells_%d" % i][0])
        assert np.array_equal(L["restrict"], agg["restrict_%d" % i])
        assert np.array_equal(L["faceRestrict"], agg["faceRestrict_%d" % i])
        assert np.array_equal(L["lowerAddr"], agg["lowerAddr_%d" % i])
        assert np.array_equal(L["upperAddr"], agg["upperAddr_%d" % i])
        assert np.array_equal(L["diag"], agg["diag_%d" % i])
        assert np.array_equal(L["upper"], agg["upper_%d" % i])
        if not S.sym:
            assert np.array_equal(L["lower"], agg["lower_%d" % i])
    ref, out = oracle.run_ref("solve", p, ds)
    x, perf = S.solve(p["psi"], p["source"], **_okw(kw))
    rp = ref["perf"]
    assert perf["nIterations"] == int(rp[2])
    assert perf["initialResidual"] == rp[0]
    assert perf["finalResidual"] == rp[1]
    assert np.array_equal(x, ref["psi"])


@pytest.mark.parametrize("gen", [lambda: cases.box3d(12), lambda: cases.box3d(9, 11, 10, asym=True),
                                 lambda: cases.random_graph(900, 5, 40), lambda: cases.jump2d(30, 30)],
                         ids=["box", "box_asym", "graph", "jump2d"])
@pytest.mark.parametrize("ncoarse", [10, 40])
def test_gamg_direct_solve_coarsest(oracle, gen, ncoarse):
    """directSolveCoarsest (GAMGSolver.C:95-106): the coarsest level by the reference's LUscalarMatrix instead of
    ICCG / BICCG - pins the oracle's dense LU (Crout, implicit scaled pivoting, LUBacksubstitute) bit for bit"""
    p = dict(gen())
    p["psi"] = np.zeros(p["nCells"])
    kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="algebraicPair", nCellsInCoarsestLevel=ncoarse,
              mergeLevels=1, cacheAgglomeration=False, tolerance=1e-9, relTol=0, directSolveCoarsest=True)
    ref, out = oracle.run_ref("solve", p, oracle.dict_string(**kw))
    okw = _okw(kw)
    okw["directSolveCoarsest"] = 1
    x, perf = oracle.System(p).solve(p["psi"], p["source"], **okw)
    rp = ref["perf"]
    assert perf["nIterations"] == int(rp[2])
    assert perf["initialResidual"] == rp[0] and perf["finalResidual"] == rp[1]
    assert np.array_equal(x, ref["psi"])


def test_gamg_preconditioned_pcg(oracle):
    p = cases.box3d(10)
    kw = dict(solver="PCG", tolerance=1e-9, relTol=0)
    sub = dict(preconditioner="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
               nCellsInCoarsestLevel=10, mergeLevels=1, cacheAgglomeration=False, nVcycles=2,
               tolerance=1e-5, relTol=0)
    ds = oracle.dict_string(**kw) + " preconditioner { " + oracle.dict_string(**sub) + " }"
    ref, out = oracle.run_ref("solve", p, ds)
    S = oracle.System(p)
    x, perf = S.solve(p["psi"], p["source"], solver="PCG", preconditioner="GAMG", smoother="GaussSeidel",
                      tolerance=1e-9, relTol=0, nVcycles=2)
    assert perf["nIterations"] == int(ref["perf"][2])
    np.testing.assert_allclose(x, ref["psi"], rtol=1e-9, atol=1e-12)


def test_nonblocking_gs_equals_gs_serially(oracle):
    """nonBlockingGaussSeidelSmoother.C: without coupled patches its cell loop is GaussSeidel's;
    the product maps the name to the GaussSeidel kernels on that ground - checked on the reference."""
    p = cases.box3d(9)
    rng = np.random.RandomState(4)
    p["psi"] = rng.randn(p["nCells"]); p["source"] = rng.randn(p["nCells"])
    kw = dict(solver="smoothSolver", nSweeps=2, tolerance=1e-12, relTol=0, maxIter=6)
    a, _ = oracle.run_ref("solve", p, oracle.dict_string(smoother="GaussSeidel", **kw))
    b, _ = oracle.run_ref("solve", p, oracle.dict_string(smoother="nonBlockingGaussSeidel", **kw))
    assert np.array_equal(a["psi"], b["psi"]) and np.array_equal(a["perf"], b["perf"])
