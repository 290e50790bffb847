"""Regression of the round-1 fuzz failure (tools/fuzz_gpu.py 900 31337, case 8723) and of the engine fallback.

The failure: the three-plane TDILU forward sweep of the cluster engine (ND = 6 instantiation) published a granule
whose second tag was overwritten by the next pointer computation - the inline 128-bit store carried one wait state,
gfx950 needs two (ldu_cluster.hip cl_store).  Consumers then waited until their spin bound: error -20.
The reference's sweeps always complete (TDILUPreconditioner.C:82-176); so must these: an operation whose
dependency wait expires is re-run on the level-kernel engine (bit-identical results)."""
import numpy as np
import pytest

from openfoam_amd import capi, cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.set_spin_limit(0)
    c.close()


@pytest.fixture(scope="module")
def fuzz8723():
    return cases.random_graph(52469, 3, 5, asym=True)


def test_fuzz_case_8723_coupled_dilu(ctx, oracle, fuzz8723):
    p = fuzz8723
    assert p["lowerAddr"].size == 68119
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    assert a.sweep_engine(0) == "clusters", "the case must run on the cluster engine"
    rng = np.random.RandomState(8723)
    S3, P3 = rng.randn(p["nCells"], 3), rng.randn(p["nCells"], 3)
    before = ctx.fallback_count()
    for rep in range(3):
        assert np.array_equal(m.coupled_precondition("DILU", S3), S.c_precondition("DILU", S3))
        assert np.array_equal(m.coupled_precondition("DILU", S3, transpose=True),
                              S.c_precondition("DILU", S3, transpose=True))
        assert np.array_equal(m.coupled_smooth(P3, S3, 2), S.c_smooth(P3, S3, 2))
        assert np.array_equal(m.precondition("DILU", S3[:, 0].copy()), S.precondition("DILU", S3[:, 0].copy())[0])
    assert ctx.fallback_count() == before, "the fixed engine must not need the fallback"
    m.close(); a.close()


@pytest.mark.parametrize("what", ["cDILU", "DILU", "GS3", "symGS", "cGS", "PBiCG", "GAMG", "cPBiCCCG"])
def test_forced_abort_falls_back_to_level_engine(ctx, oracle, fuzz8723, what):
    """spin bound of ONE poll: every point-to-point / cluster sweep with a dependency outside its wave aborts;
    the operation must still return what the reference computes."""
    big = what in ("cDILU", "DILU", "GS3", "symGS", "cGS")
    # (PBiCG amplifies the summation order of its global sums: a moderate size and tolerance keep the histories
    #  comparable, as in test_gpu_parity.py)
    p = fuzz8723 if big else cases.box3d(30, 28, 26, asym=(what != "GAMG"))
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    n = p["nCells"]
    rng = np.random.RandomState(5)
    v, w = rng.randn(n), rng.randn(n)
    V3, W3 = rng.randn(n, 3), rng.randn(n, 3)
    ctx.set_spin_limit(1)
    before = ctx.fallback_count()
    try:
        if what == "cDILU":
            assert np.array_equal(m.coupled_precondition("DILU", V3), S.c_precondition("DILU", V3))
        elif what == "DILU":
            assert np.array_equal(m.precondition("DILU", v), S.precondition("DILU", v)[0])
        elif what == "GS3":
            assert np.array_equal(m.smooth("GaussSeidel", v, w, 3), S.smooth("GaussSeidel", v, w, 3))
        elif what == "symGS":
            assert np.array_equal(m.smooth("symGaussSeidel", v, w, 2), S.smooth("symGaussSeidel", v, w, 2))
        elif what == "cGS":
            assert np.array_equal(m.coupled_smooth(V3, W3, 2), S.c_smooth(V3, W3, 2))
        elif what == "PBiCG":
            kw = dict(tolerance=1e-6, relTol=0)
            x, perf = m.solve(p["psi"], p["source"], solver="PBiCG", preconditioner="DILU", **kw)
            xo, po = S.solve(p["psi"], p["source"], solver="PBiCG", precond="DILU", **kw)
            assert perf["nIterations"] == po["nIterations"]
            np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-5, atol=1e-9)
            assert np.max(np.abs(x - xo)) <= 1e-6 * np.max(np.abs(xo))
        elif what == "GAMG":
            kw = dict(tolerance=1e-8, relTol=0)
            x, perf = m.solve(p["psi"], p["source"], solver="GAMG", smoother="GaussSeidel", **kw)
            xo, po = S.solve(p["psi"], p["source"], solver="GAMG", smoother="GaussSeidel", **kw)
            assert perf["nIterations"] == po["nIterations"]
            np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-6, atol=1e-12)
            assert np.max(np.abs(x - xo)) <= 1e-8 * np.max(np.abs(xo))
        else:
            x, perf = m.coupled_solve(V3 * 0, W3, solver="PBiCCCG", preconditioner="DILU", tolerance=1e-8, relTol=0)
            xo, po = S.c_solve(V3 * 0, W3, solver="PBiCCCG", preconditioner="DILU", tolerance=1e-8, relTol=0)
            assert perf["nIterations"] == po["nIterations"]
            assert np.max(np.abs(x - xo)) <= 1e-6 * np.max(np.abs(xo))
        assert ctx.fallback_count() > before, "the forced abort did not happen: the test proves nothing"
    finally:
        ctx.set_spin_limit(0)
    # and the fast engines are back afterwards
    before = ctx.fallback_count()
    assert np.array_equal(m.smooth("GaussSeidel", v, w, 2), S.smooth("GaussSeidel", v, w, 2))
    assert ctx.fallback_count() == before
    m.close(); a.close()


@pytest.mark.parametrize("what", ["GS2_cluster", "DIC_cluster", "GS3_level_engine", "GS4_block_engine", "GAMG"])
def test_crawling_sweep_trips_the_time_watchdog(ctx, oracle, what):
    """the spin bound counts polls, so a launch that merely CRAWLS never reached it.  Every wait is
    now bounded in wall-clock time as well (ldu_ctx_set_watchdog).  Injected here: the wave that runs the first task of
    each sweep launch stalls for 60 ms under a 10 ms budget - every wave behind it gives up, the operation is re-run on
    the level-kernel engine and must still return what the reference computes."""
    p = cases.box3d(44) if what not in ("GS3_level_engine", "GS4_block_engine") else cases.random_graph(30000, 9 if what == "GS4_block_engine" else 4, 60)
    S = oracle.System(p)
    a, m = capi.from_problem(ctx, p)
    if what.endswith("cluster"):
        assert a.sweep_engine(0) == "clusters"
    n = p["nCells"]
    rng = np.random.RandomState(11)
    v, w = rng.randn(n), rng.randn(n)
    before = ctx.fallback_count()
    ctx.set_watchdog(10.0, 60.0)
    try:
        if what == "GS2_cluster":
            assert np.array_equal(m.smooth("GaussSeidel", v, w, 2), S.smooth("GaussSeidel", v, w, 2))
        elif what == "DIC_cluster":
            assert np.array_equal(m.precondition("DIC", v), S.precondition("DIC", v)[0])
        elif what == "GS3_level_engine":
            assert np.array_equal(m.smooth("GaussSeidel", v, w, 3), S.smooth("GaussSeidel", v, w, 3))
        elif what == "GS4_block_engine":
            assert a.sweep_engine(2) == "blocks"
            assert np.array_equal(m.smooth("GaussSeidel", v, w, 4), S.smooth("GaussSeidel", v, w, 4))
        else:
            kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
                      mergeLevels=1, tolerance=1e-7, relTol=0.01)
            x, perf = m.solve(p["psi"], p["source"], cacheAgglomeration=1, **kw)
            xo, po = S.solve(p["psi"], p["source"], **kw)
            assert perf["nIterations"] == po["nIterations"]
            np.testing.assert_allclose(perf["history"], po["history"], rtol=1e-6, atol=1e-12)
        assert ctx.fallback_count() > before, "the stalled launch must have tripped the watchdog"
    finally:
        ctx.set_watchdog(200.0, 0.0)
    # the fast engines are back, and quiet
    before = ctx.fallback_count()
    assert np.array_equal(m.smooth("GaussSeidel", v, w, 2), S.smooth("GaussSeidel", v, w, 2))
    assert ctx.fallback_count() == before
    m.close(); a.close()
