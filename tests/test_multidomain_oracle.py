"""Multi-rank semantics of the CPU oracle's serial emulation (no GPU): the decomposed system
(processor patches, rank-local preconditioners / smoothers / agglomeration) against the
undecomposed one wherever the algorithm is decomposition-invariant, and self-consistency
elsewhere.  The reference cannot pin this here (no MPI in the container): SURVEY.md 8c."""
import numpy as np
import pytest

from openfoam_amd import cases, decompose


def _split(p, n, kind="slab"):
    nx = round(p["nCells"] ** (1.0 / 3))
    if kind == "slab":
        cr = decompose.slab_ranks(nx, nx, nx, n)
    else:
        cr = decompose.block_ranks(nx, nx, nx, 2, 2, 2 if n == 8 else 1)
    return decompose.decompose(p, cr, n)


@pytest.mark.parametrize("asym", [False, True])
@pytest.mark.parametrize("n", [2, 4])
def test_ops_match_undecomposed(oracle, asym, n):
    p = cases.box3d(8, asym=asym)
    rng = np.random.RandomState(3)
    x = rng.randn(p["nCells"]); b = rng.randn(p["nCells"])
    subs, maps = _split(p, n, "slab" if n == 2 else "block")
    S1 = oracle.System(p)
    SN = oracle.System(subs)
    xs = np.concatenate([x[m] for m in maps]); bs = np.concatenate([b[m] for m in maps])
    perm = np.concatenate(maps)
    for name in ("Amul", "Tmul"):
        y1 = getattr(S1, name)(x)
        yN = getattr(SN, name)(xs)
        np.testing.assert_allclose(yN, y1[perm], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(SN.sumA(), S1.sumA()[perm], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(SN.residual(xs, bs), S1.residual(x, b)[perm], rtol=1e-13, atol=1e-12)


def test_pcg_diagonal_is_decomposition_invariant(oracle):
    p = cases.box3d(8)
    subs, maps = _split(p, 2)
    kw = dict(solver="PCG", precond="diagonal", tolerance=1e-9, relTol=0)
    x1, p1 = oracle.System(p).solve(p["psi"], p["source"], **kw)
    xs = np.concatenate([p["psi"][m] for m in maps]); bs = np.concatenate([p["source"][m] for m in maps])
    xN, pN = oracle.System(subs).solve(xs, bs, **kw)
    assert pN["nIterations"] == p1["nIterations"]
    np.testing.assert_allclose(pN["history"], p1["history"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(xN, x1[np.concatenate(maps)], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("kw", [
    dict(solver="PCG", precond="DIC", tolerance=1e-9, relTol=0),
    dict(solver="GAMG", smoother="GaussSeidel", tolerance=1e-9, relTol=0, nCellsInCoarsestLevel=4),
    dict(solver="GAMG", smoother="GaussSeidel", tolerance=1e-9, relTol=0, nCellsInCoarsestLevel=4,
         mergeLevels=2, nPreSweeps=1),
])
def test_decomposed_solvers_converge_to_same_solution(oracle, kw):
    p = cases.box3d(10)
    subs, maps = _split(p, 4, "block")
    xs = np.concatenate([p["psi"][m] for m in maps]); bs = np.concatenate([p["source"][m] for m in maps])
    xN, pN = oracle.System(subs).solve(xs, bs, **kw)
    assert pN["converged"]
    x1, p1 = oracle.System(p).solve(p["psi"], p["source"], **dict(kw, tolerance=1e-11))
    np.testing.assert_allclose(xN, x1[np.concatenate(maps)], rtol=2e-5, atol=2e-6)
    # block-local preconditioning costs iterations, never fewer than ~the serial count / 2
    assert pN["nIterations"] >= 1


def test_openmp_build_is_bit_identical(tmp_path):
    """oracle/libldu_oracle_omp.so (one thread per sub-domain = per emulated rank; the all-host-cores CPU baseline of
    bench.py) must reproduce the serial emulation bit for bit: per-domain loops are untouched, sums stay in rank order."""
    import os
    import subprocess
    import sys
    import numpy as np
    from openfoam_amd import cases, decompose
    import oracle_py
    here = os.path.dirname(os.path.abspath(__file__))
    script = tmp_path / "run.py"
    script.write_text('''
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import conftest
from openfoam_amd import cases, decompose
import oracle_py
p = cases.box3d(12, asym=(sys.argv[1] == "asym"))
subs, _ = decompose.decompose(p, decompose.block_ranks(12, 12, 12, 2, 2, 1), 4)
S = oracle_py.System(subs)
X0 = np.concatenate([s["psi"] for s in subs]); B = np.concatenate([s["source"] for s in subs])
out = {}
kw = dict(tolerance=1e-9, relTol=0)
if sys.argv[1] == "asym":
    x, pf = S.solve(X0, B, solver="PBiCG", precond="DILU", **kw)
else:
    x, pf = S.solve(X0, B, solver="PCG", precond="DIC", **kw)
out["k_x"], out["k_h"] = x, pf["history"]
x, pf = S.solve(X0, B, solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
                mergeLevels=1, **kw)
out["g_x"], out["g_h"] = x, pf["history"]
out["gs"] = S.smooth("GaussSeidel", X0 + 1.0, B, 3)
out["amul"] = S.Amul(B)
np.savez(sys.argv[2], **out)
''' % (here, os.path.join(here, "..", "oracle")))
    for kind in ("sym", "asym"):
        res = {}
        for omp in ("0", "1"):
            f = str(tmp_path / ("out_%s_%s.npz" % (kind, omp)))
            env = dict(os.environ, LDU_ORACLE_OMP=omp, OMP_NUM_THREADS="4")
            subprocess.run([sys.executable, str(script), kind, f], check=True, env=env)
            res[omp] = dict(np.load(f))
        for k in res["0"]:
            assert np.array_equal(res["0"][k], res["1"][k]), (kind, k)
