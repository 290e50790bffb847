"""GPU (-m gpu): every sweep engine variant must reproduce the oracle's sequential sweeps bit for bit:
the cluster (row-blocking) engine, the single-wavefront LDS kernel of small matrices, the chip-wide point-to-point engine
(LDU_P2P_SLABS=0), the XCD-slab engine with 1, 3 and 8 slabs
(cross-slab dependencies through the write-through copies), pipelined GaussSeidel on both, and the
level-kernel engine.  Cases: hex box (regular DAG), asymmetric box, irregular graph with wide rows."""
import os

import numpy as np
import pytest

from openfoam_amd import capi, cases

pytestmark = pytest.mark.gpu

ENGINES = {
    "chip": {"LDU_P2P_SLABS": "0"},
    "slab1": {"LDU_P2P_SLABS": "1"},
    "slab3": {"LDU_P2P_SLABS": "3"},
    "slab8": {"LDU_P2P_SLABS": "8"},
    "slab8_bpc3": {"LDU_P2P_SLABS": "8", "LDU_P2P_BPC": "3"},
    "auto": {},
    "levels": {"LDU_SWEEP": "levels"},
    "nosmall": {"LDU_SMALL": "0"},
    "cluster": {"LDU_CLUSTER": "2", "LDU_CLUSTER_MIN": "1"},
    "cluster_bpc1": {"LDU_CLUSTER": "2", "LDU_CLUSTER_MIN": "1", "LDU_CLUSTER_BPC": "1"},
    "nocluster": {"LDU_CLUSTER": "0"},
    "small8192": {"LDU_SMALL_MAX": "8192"},
}
KEYS = ("LDU_P2P_SLABS", "LDU_P2P_BPC", "LDU_SWEEP", "LDU_SMALL", "LDU_SMALL_MAX", "LDU_CLUSTER", "LDU_CLUSTER_MIN",
        "LDU_CLUSTER_BPC")


def _problems():
    rng = np.random.RandomState(11)
    out = {}
    p = cases.box3d(24, 20, 33)
    p["psi"] = rng.randn(p["nCells"])
    out["box"] = p
    p = cases.box3d(17, 30, 21, asym=True)
    p["psi"] = rng.randn(p["nCells"])
    out["box_asym"] = p
    p = cases.random_graph(30000, 9, 400)
    p["psi"] = rng.randn(p["nCells"])
    out["graph"] = p
    # <= 8192 cells: the single-wavefront LDS kernel (engine "small8192"; default limit 3000 cells: "chain")
    p = cases.box3d(19, 20, 21)
    p["psi"] = rng.randn(p["nCells"])
    out["box_small"] = p
    p = cases.random_graph(5000, 11, 300, asym=True)   # rows wider than the 8-entry fast path
    p["psi"] = rng.randn(p["nCells"])
    out["graph_small"] = p
    p = cases.random_graph(20000, 2, 60)   # irregular, <= 6 lower / upper neighbours: cluster-eligible
    p["psi"] = rng.randn(p["nCells"])
    out["graph_sparse"] = p
    p = cases.laplacian2d(1, 37)   # a chain: one row per level
    p["psi"] = rng.randn(p["nCells"])
    out["chain"] = p
    return out


@pytest.fixture(scope="module")
def expected(oracle):
    exp = {}
    for name, p in _problems().items():
        S = oracle.System(p)
        src, psi = p["source"], p["psi"]
        sym = "lower" not in p
        e = dict(p=p)
        e["gs1"] = S.smooth("GaussSeidel", psi, src, 1)
        e["gs4"] = S.smooth("GaussSeidel", psi, src, 4)
        e["sgs2"] = S.smooth("symGaussSeidel", psi, src, 2)
        if sym:
            e["dic"] = S.precondition("DIC", src)[0]
            e["dicgs"] = S.smooth("DICGaussSeidel", psi, src, 2)
        else:
            e["dilu"] = S.precondition("DILU", src)[0]
            e["diluT"] = S.precondition("DILU", src, transpose=True)[0]
            e["dilugs"] = S.smooth("DILUGaussSeidel", psi, src, 2)
        exp[name] = e
    return exp


@pytest.mark.parametrize("engine", sorted(ENGINES))
def test_sweeps_bitexact_on_every_engine(engine, expected):
    saved = {k: os.environ.pop(k, None) for k in KEYS}
    os.environ.update(ENGINES[engine])
    try:
        ctx = capi.Context(0)
        for name, e in expected.items():
            p = e["p"]
            a, m = capi.from_problem(ctx, p)
            src, psi = p["source"], p["psi"]
            for rep in range(2):   # second pass: epochs, ticket parity and cached factors
                assert np.array_equal(m.smooth("GaussSeidel", psi, src, 1), e["gs1"]), (engine, name, "gs1")
                assert np.array_equal(m.smooth("GaussSeidel", psi, src, 4), e["gs4"]), (engine, name, "gs4")
                assert np.array_equal(m.smooth("symGaussSeidel", psi, src, 2), e["sgs2"]), (engine, name, "sgs2")
                if "dic" in e:
                    assert np.array_equal(m.precondition("DIC", src), e["dic"]), (engine, name, "dic")
                    assert np.array_equal(m.smooth("DICGaussSeidel", psi, src, 2), e["dicgs"]), (engine, name)
                else:
                    assert np.array_equal(m.precondition("DILU", src), e["dilu"]), (engine, name, "dilu")
                    assert np.array_equal(m.precondition("DILU", src, transpose=True), e["diluT"]), (engine, name)
                    assert np.array_equal(m.smooth("DILUGaussSeidel", psi, src, 2), e["dilugs"]), (engine, name)
            m.close(); a.close()
        ctx.close()
    finally:
        for k in KEYS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
