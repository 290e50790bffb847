"""GPU (-m gpu): sub-domain mode (include/ldugpu.h: ldu_addr_set_subdomains) - K ranks of the reference
inside ONE addressing and one set of launches.  The K sub-domains of a decomposition are handed over concatenated, every
processor patch as a cyclic patch paired with its counterpart (decompose.concatenate); every operator is then the K-rank
operator of the reference (GaussSeidelSmoother.C:98-145: the neighbour ranks' values of the previous sweep;
lduMatrixUpdateMatrixInterfaces.C:30-160), and GAMG coarsens rank by rank (pairs never cross an interface; the
and-reduce of continueAgglomerating, GAMGAgglomeration.C:53-62).  Checker: the multi-domain oracle at the same K
(oracle_py.System(subs): the K-rank algorithm, its coarse-interface ordering pinned on reference runs, row a31) - Amul,
residual and GaussSeidel sweeps bit for bit per sub-domain, the GAMG hierarchy level by level, histories to 1e-6."""
import numpy as np
import pytest

from openfoam_amd import capi, cases, decompose

pytestmark = pytest.mark.gpu
GAMG = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10, mergeLevels=1,
            tolerance=1e-7, relTol=0.01)


def _renumber(p):
    order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
    nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
    return cases.renumbered(p, order, fmap, flip, nl, nu)


def _cases():
    out = []
    p = cases.box3d(24)
    out.append(("box24_2x2x2", p, decompose.block_ranks(24, 24, 24, 2, 2, 2), 8))
    p = _renumber(cases.irregular_box(30))
    out.append(("irregular30_blobs5", p, None, 5))
    p = cases.box3d(40, 36, 44)
    out.append(("box_blobs32", p, None, 32))
    return out


@pytest.mark.parametrize("env", [{}, {"LDU_BLK_MIN": "1", "LDU_WG": "0", "LDU_SMALL": "0", "LDU_BLK_CELLS": "700"}, {"LDU_BLK_IFACE": "0"}],
                         ids=["default", "blocks_everywhere", "level_engines"])
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c[0])
def test_subdomain_mode_against_the_multidomain_oracle(oracle, case, env, monkeypatch):
    """(env: the levels with interfaces on the block engine - pipelined sweeps, interface values by sweep parity - where it
    applies / on every level down to the coarsest / nowhere: the level engines sweep by sweep)"""
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    name, p, rank, K = case
    if rank is None:
        rank = decompose.blob_ranks(p["nCells"], p["lowerAddr"], p["upperAddr"], K)
        K = int(rank.max()) + 1
    assert np.bincount(rank, minlength=K).min() > 50
    subs, maps = decompose.decompose(p, rank, K)
    cp = decompose.concatenate(subs)
    S = oracle.System(subs)
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, cp)
    rng = np.random.RandomState(8)
    x, b = rng.randn(cp["nCells"]), rng.randn(cp["nCells"])
    assert np.array_equal(m.Amul(x), S.Amul(x))
    assert np.array_equal(m.residual(x, b), S.residual(x, b))
    for k in (1, 2, 4):
        assert np.array_equal(m.smooth("GaussSeidel", x, b, k), S.smooth("GaussSeidel", x, b, k)), k
    xg, pg = m.solve(cp["psi"], cp["source"], **GAMG)
    xo, po = S.solve(cp["psi"], cp["source"], **GAMG)
    assert pg["nIterations"] == po["nIterations"], (pg["nIterations"], po["nIterations"])
    np.testing.assert_allclose(pg["history"], po["history"], rtol=1e-6, atol=1e-12)
    assert np.max(np.abs(xg - xo)) <= 1e-7 * np.max(np.abs(xo))
    # the hierarchy is the K-rank hierarchy: as many levels, as many cells per level as the oracle's K hierarchies together
    lv = m.gamg_level_sizes(**GAMG)
    lo = S.gamg_level_sizes(**GAMG)
    assert len(lv) == len(lo), (len(lv), len(lo))
    assert [(L["nCells"], L["nFaces"]) for L in lv] == lo, ([(L["nCells"], L["nFaces"]) for L in lv], lo)
    engines = [a.sweep_engine(2)] + [L["engine_gs_multi"] for L in lv]
    if env.get("LDU_BLK_IFACE") == "0":
        assert "blocks" not in engines, engines
    elif env or p["nCells"] >= 2000:
        assert engines[0] == "blocks", engines
    assert ctx.fallback_count() == 0
    # and the one-domain solve of the same matrix needs no more V-cycles than the K-rank algorithm (sanity of the comparison)
    a1, m1 = capi.from_problem(ctx, p)
    x1, p1 = m1.solve(p["psi"], p["source"], **GAMG)
    assert p1["nIterations"] <= pg["nIterations"] + 1
    print("%s: K = %d, %d levels, %d V-cycles (one domain: %d), engines %s" % (name, K, len(lv), pg["nIterations"], p1["nIterations"], engines))
    m1.close(); a1.close(); m.close(); a.close(); ctx.close()


def test_subdomain_mode_on_the_real_mesh(oracle):
    """the 1.73 M-cell snappyHexMesh motorBike mesh (bandCompression numbering) cut into 8 sub-domains: GaussSeidel bit for bit
    against the 8-rank oracle, the GAMG solve by history; and what the mode is for - the dependency DAG of every level is
    shallower than the one-domain DAG of the same level"""
    from openfoam_amd import motorbike
    if not motorbike.available("mb2"):
        pytest.skip("data/motorbike/mb2.npz not present")
    p = motorbike.problem("mb2")
    p.pop("cellLevel"); p.pop("meta")
    p = _renumber(p)
    rank = decompose.blob_ranks(p["nCells"], p["lowerAddr"], p["upperAddr"], 8)
    K = int(rank.max()) + 1
    subs, maps = decompose.decompose(p, rank, K)
    cp = decompose.concatenate(subs)
    cq, order = decompose.concatenated(p, rank, K)
    assert np.array_equal(cp["lowerAddr"], cq["lowerAddr"]) and np.array_equal(cp["diag"], cq["diag"]) and len(cp["patches"]) == len(cq["patches"])
    S = oracle.System(subs)
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, cp)
    rng = np.random.RandomState(9)
    x, b = rng.randn(cp["nCells"]), rng.randn(cp["nCells"])
    assert np.array_equal(m.Amul(x), S.Amul(x))
    for k in (1, 4):
        assert np.array_equal(m.smooth("GaussSeidel", x, b, k), S.smooth("GaussSeidel", x, b, k)), k
    xg, pg = m.solve(cp["psi"], cp["source"], **GAMG)
    xo, po = S.solve(cp["psi"], cp["source"], **GAMG)
    assert pg["nIterations"] == po["nIterations"]
    np.testing.assert_allclose(pg["history"], po["history"], rtol=1e-6, atol=1e-12)
    a1, m1 = capi.from_problem(ctx, p)
    x1, p1 = m1.solve(p["psi"], p["source"], **GAMG)
    d8 = [a.info()["nLevels"]] + [L["nLevels"] for L in m.gamg_level_sizes(**GAMG)]
    d1 = [a1.info()["nLevels"]] + [L["nLevels"] for L in m1.gamg_level_sizes(**GAMG)]
    print("mb2, 8 sub-domains: %d V-cycles (one domain: %d); dependency levels per GAMG level %s (one domain: %s)"
          % (pg["nIterations"], p1["nIterations"], d8, d1))
    assert all(x8 <= x1_ for x8, x1_ in zip(d8[:6], d1[:6]))
    assert ctx.fallback_count() == 0
    m1.close(); a1.close(); m.close(); a.close(); ctx.close()
