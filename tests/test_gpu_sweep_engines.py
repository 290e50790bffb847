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
    "chip": {"LDU_BLK": "0", "LDU_P2P_SLABS": "0"},
    "slab1": {"LDU_BLK": "0", "LDU_P2P_SLABS": "1"},
    "slab3": {"LDU_BLK": "0", "LDU_P2P_SLABS": "3"},
    "slab8": {"LDU_BLK": "0", "LDU_P2P_SLABS": "8"},
    "slab8_bpc3": {"LDU_BLK": "0", "LDU_P2P_SLABS": "8", "LDU_P2P_BPC": "3"},
    "auto": {},
    "levels": {"LDU_SWEEP": "levels"},
    "nosmall": {"LDU_BLK": "0", "LDU_SMALL": "0"},
    "cluster": {"LDU_CLUSTER": "2", "LDU_CLUSTER_MIN": "1"},
    "cluster_bpc1": {"LDU_CLUSTER": "2", "LDU_CLUSTER_MIN": "1", "LDU_CLUSTER_BPC": "1"},
    "nocluster": {"LDU_BLK": "0", "LDU_CLUSTER": "0"},
    "small8192": {"LDU_BLK": "0", "LDU_SMALL_MAX": "8192"},
    "small16384": {"LDU_BLK": "0", "LDU_SMALL_MAX": "16384"},
    "small_nopipe": {"LDU_BLK": "0", "LDU_SMALL_PIPE": "0"},
    "nocoop": {"LDU_BLK": "0", "LDU_COOP_ROWS": "0"},
    "nolag": {"LDU_BLK": "0", "LDU_LAG_BUCKETS": "0"},
    "nosort": {"LDU_BLK": "0", "LDU_SORT_ROWS": "0"},
    "nowg": {"LDU_BLK": "0", "LDU_WG": "0"},
    "wg4": {"LDU_BLK": "0", "LDU_WG_WAVES": "4"},
    "nowg_small16384": {"LDU_BLK": "0", "LDU_WG": "0", "LDU_SMALL_MAX": "16384"},
    "wg_wide18000": {"LDU_BLK": "0", "LDU_WG_MAX": "18000", "LDU_WG_WIDE": "1"},
    "cluster_via_level_layout": {"LDU_CLUSTER": "2", "LDU_CLUSTER_MIN": "1", "LDU_CLUSTER_DIRECT": "0"},
    # LDS-resident blocks (ldu_blocks.hip) forced onto the small test matrices: many small blocks / a few large ones,
    # seven or three compute wavefronts per block
    "blocks": {"LDU_CLUSTER": "0", "LDU_BLK_MIN": "1", "LDU_BLK_CELLS": "400", "LDU_WG": "0", "LDU_SMALL": "0"},
    "blocks_w3": {"LDU_CLUSTER": "0", "LDU_BLK_MIN": "1", "LDU_BLK_CELLS": "1500", "LDU_BLK_WAVES": "3", "LDU_WG": "0", "LDU_SMALL": "0"},
    "blocks_auto": {"LDU_CLUSTER": "0", "LDU_BLK_MIN": "1", "LDU_WG": "0", "LDU_SMALL": "0"},
    "noblocks": {"LDU_BLK": "0"},
}
KEYS = ("LDU_P2P_SLABS", "LDU_P2P_BPC", "LDU_SWEEP", "LDU_SMALL", "LDU_SMALL_MAX", "LDU_CLUSTER", "LDU_CLUSTER_MIN",
        "LDU_CLUSTER_BPC", "LDU_SMALL_PIPE", "LDU_COOP_ROWS", "LDU_SORT_ROWS", "LDU_LAG_BUCKETS", "LDU_WG", "LDU_WG_WAVES",
        "LDU_WG_MAX", "LDU_WG_MIN", "LDU_WG_WIDE", "LDU_CLUSTER_DIRECT", "LDU_BLK", "LDU_BLK_MIN", "LDU_BLK_MAX",
        "LDU_BLK_CELLS", "LDU_BLK_WAVES", "LDU_BLK_CELLS_MIN", "LDU_BLK_CELLS_MAX", "LDU_BLK_PER_CU", "LDU_BLK_WPS",
        "LDU_BLK_XCD", "LDU_BLK_LAYOUTS", "LDU_GS_LAYOUTS", "LDU_GS_LAYOUTS_MIN")


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
    # rows with up to ~40 lower / upper neighbours: the cooperative rows of the level engines (2 / 4 / 8 lanes per row)
    p = cases.random_graph(12000, 30, 500, asym=True)
    p["psi"] = rng.randn(p["nCells"])
    out["graph_wide"] = p
    p = cases.random_graph(9000, 22, 300)
    p["psi"] = rng.randn(p["nCells"])
    out["graph_wide_sym"] = p
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


@pytest.mark.parametrize("pipe", ["1", "0"])
def test_small_pipelined_sweeps_bitexact(oracle, pipe):
    """gs_small_pipe_kernel: k GaussSeidel sweeps of a small matrix pipelined over k wavefronts of ONE workgroup
    (sweep j trails sweep j-1 by the upper-neighbour lag, one solution vector in LDS) - every k the GAMG V-cycle can
    ask for (2 ... 8: chunks of 4 / 3 / 2), tiny to 6000 cells, narrow and wide rows, hex, random and chain graphs;
    bit-exact against the sequential sweeps; LDU_SMALL_PIPE=0 = the single-wavefront kernel."""
    saved = {k: os.environ.pop(k, None) for k in KEYS}
    os.environ["LDU_SMALL_PIPE"] = pipe
    os.environ["LDU_WG"] = "0"
    os.environ["LDU_BLK"] = "0"
    try:
        ctx = capi.Context(0)
        rng = np.random.RandomState(4)
        probs = [cases.box3d(3, 4, 3), cases.box3d(11, 12, 13), cases.box3d(18, 18, 18), cases.random_graph(2999, 9, 200),
                 cases.random_graph(5900, 5, 150, asym=True), cases.random_graph(700, 13, 60), cases.laplacian2d(1, 50),
                 cases.laplacian2d(64, 64), cases.irregular_box(17)]
        for p in probs:
            a, m = capi.from_problem(ctx, p)
            # (rows wider than 16 entries stay on the chip-wide engines: the comparison below still holds)
            if p["nCells"] in (36, 1716) or (pipe == "1" and p["nCells"] == 5832):
                assert a.sweep_engine(2) == "single wavefront", p["nCells"]
            S = oracle.System(p)
            psi, src = rng.randn(p["nCells"]), rng.randn(p["nCells"])
            for k in (2, 3, 4, 5, 6, 7, 8):
                for rep in range(2):
                    assert np.array_equal(m.smooth("GaussSeidel", psi, src, k), S.smooth("GaussSeidel", psi, src, k)), (p["nCells"], k)
            m.close(); a.close()
        ctx.close()
    finally:
        for k in KEYS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.parametrize("waves", ["8", "4"])
def test_workgroup_engine_bitexact(oracle, waves):
    """gs_wg_kernel: k = 1 ... 8 GaussSeidel sweeps of a matrix of up to 20 000 cells as LDS-synchronised (sweep, slice)
    tasks of ONE workgroup (solution vector in LDS, level counters); tiny to 19 683 cells, rows of up to ~60 entries (the
    chunked tail), hex / random / chain / octree-coarse-like graphs; bit-exact against the sequential sweeps."""
    saved = {k: os.environ.pop(k, None) for k in KEYS}
    os.environ["LDU_WG_WAVES"] = waves
    os.environ["LDU_WG_MAX"] = "18000"
    os.environ["LDU_WG_WIDE"] = "1"
    os.environ["LDU_BLK"] = "0"
    try:
        ctx = capi.Context(0)
        rng = np.random.RandomState(5)
        probs = [cases.box3d(3, 4, 3), cases.box3d(1, 1, 2), cases.box3d(11, 12, 13), cases.box3d(26, 26, 26),
                 cases.random_graph(2999, 9, 200), cases.random_graph(5900, 5, 150, asym=True), cases.random_graph(700, 13, 60),
                 cases.laplacian2d(1, 50), cases.laplacian2d(100, 100), cases.irregular_box(25),
                 cases.random_graph(12000, 30, 500, asym=True), cases.random_graph(17900, 12, 3000)]
        for p in probs:
            a, m = capi.from_problem(ctx, p)
            assert a.sweep_engine(2) == "one workgroup", p["nCells"]
            S = oracle.System(p)
            psi, src = rng.randn(p["nCells"]), rng.randn(p["nCells"])
            for k in (1, 2, 3, 4, 5, 6, 7, 8):
                for rep in range(2):
                    assert np.array_equal(m.smooth("GaussSeidel", psi, src, k), S.smooth("GaussSeidel", psi, src, k)), (p["nCells"], k)
            assert ctx.fallback_count() == 0
            m.close(); a.close()
        ctx.close()
    finally:
        for k in KEYS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.parametrize("waves,cells,extra", [("7", "0", {}), ("7", "300", {}), ("3", "900", {}), ("7", "5000", {}),
                                               ("7", "700", {"LDU_BLK_WPS": "2", "LDU_BLK_XCD": "0"}),
                                               ("7", "500", {"LDU_BLK_LAYOUTS": "2"})])
def test_block_engine_bitexact(oracle, waves, cells, extra):
    """gs_blk_kernel (ldu_blocks.hip): k = 1 ... 8 GaussSeidel sweeps with the matrix cut into blocks that live in the LDS of
    one workgroup each (values + sweep stamps; granules and an importer wavefront between blocks); hex / random / chain /
    wide-row graphs (rows of up to ~60 entries: the chunked tail), one block ... a few hundred blocks; bit-exact against
    the sequential sweeps, no engine fallback."""
    saved = {k: os.environ.pop(k, None) for k in KEYS}
    os.environ.update({"LDU_BLK_MIN": "1", "LDU_BLK_WAVES": waves, "LDU_BLK_CELLS": cells, "LDU_WG": "0", "LDU_SMALL": "0",
                       "LDU_CLUSTER": "0"})
    os.environ.update(extra)
    try:
        ctx = capi.Context(0)
        rng = np.random.RandomState(6)
        probs = [cases.box3d(3, 4, 3), cases.box3d(11, 12, 13), cases.box3d(26, 26, 26), cases.box3d(40, 37, 41),
                 cases.random_graph(2999, 9, 200), cases.random_graph(5900, 5, 150, asym=True), cases.random_graph(700, 13, 60),
                 cases.laplacian2d(1, 50), cases.laplacian2d(100, 100), cases.irregular_box(25), cases.irregular_box(40),
                 cases.random_graph(12000, 30, 500, asym=True), cases.random_graph(17900, 12, 3000),
                 cases.random_graph(60000, 7, 900)]
        for p in probs:
            a, m = capi.from_problem(ctx, p)
            S = oracle.System(p)
            psi, src = rng.randn(p["nCells"]), rng.randn(p["nCells"])
            for k in (1, 2, 3, 4, 5, 6, 7, 8):
                for rep in range(2):
                    assert np.array_equal(m.smooth("GaussSeidel", psi, src, k), S.smooth("GaussSeidel", psi, src, k)), (p["nCells"], k)
            if p["nCells"] > 100 and not (cells == "300" and p["nCells"] > 60000):
                assert a.sweep_engine(2) == "blocks", p["nCells"]
            assert ctx.fallback_count() == 0
            m.close(); a.close()
        ctx.close()
    finally:
        for k in KEYS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.parametrize("extra", [{}, {"LDU_COOP_ROWS": "0"}, {"LDU_P2P_BPC": "1"}], ids=["default", "nocoop", "bpc1"])
def test_gs_layouts_bitexact(oracle, extra):
    """Per-sweep layouts of the chip-wide pipelined GaussSeidel sweeps (ldu_gslayouts.cpp): sweep j >= 1 of a launch runs on its
    own slices (rows of equal time T_j in the row DAG of the k sweeps, own entry tables, rows addressed through rowIdx); forced
    onto small matrices (LDU_GS_LAYOUTS_MIN=1, every other engine off); hex / random / chain / wide-row graphs, k = 1 ... 8,
    coefficient changes on the same matrix; bit-exact against the sequential sweeps, no engine fallback."""
    saved = {k: os.environ.pop(k, None) for k in KEYS}
    os.environ.update({"LDU_BLK": "0", "LDU_CLUSTER": "0", "LDU_WG": "0", "LDU_SMALL": "0", "LDU_P2P_SLABS": "0",
                       "LDU_GS_LAYOUTS": "1", "LDU_GS_LAYOUTS_MIN": "1"})
    os.environ.update(extra)
    try:
        ctx = capi.Context(0)
        rng = np.random.RandomState(7)
        probs = [cases.box3d(3, 4, 3), cases.box3d(11, 12, 13), cases.box3d(40, 37, 41), cases.random_graph(2999, 9, 200),
                 cases.random_graph(5900, 5, 150, asym=True), cases.random_graph(700, 13, 60), cases.laplacian2d(1, 50),
                 cases.laplacian2d(100, 100), cases.irregular_box(40), cases.random_graph(12000, 30, 500, asym=True),
                 cases.random_graph(17900, 12, 3000), cases.random_graph(60000, 7, 900), cases.random_graph(200, 90, 150)]
        for p in probs:
            a, m = capi.from_problem(ctx, p)
            S = oracle.System(p)
            psi, src = rng.randn(p["nCells"]), rng.randn(p["nCells"])
            for k in (1, 2, 3, 4, 5, 6, 7, 8):
                for rep in range(2):
                    assert np.array_equal(m.smooth("GaussSeidel", psi, src, k), S.smooth("GaussSeidel", psi, src, k)), (p["nCells"], k)
            assert a.sweep_engine(2) == "chip-wide point-to-point", p["nCells"]
            lay = m.gs_layouts()
            assert lay["built"] == 4 and all(v > 0 for v in lay["slices"]), (p["nCells"], lay)
            # new coefficients on the same matrix object: the layouts' value arrays follow
            q = dict(p)
            q["diag"] = p["diag"] * (1.0 + 0.25 * rng.rand(p["nCells"]))
            q["upper"] = p["upper"] * (1.0 + 0.25 * rng.rand(p["upper"].size))
            if "lower" in p:
                q["lower"] = p["lower"] * (1.0 + 0.25 * rng.rand(p["upper"].size))
            m.set_coeffs(q["diag"], q["upper"], q.get("lower"))
            S2 = oracle.System(q)
            for k in (2, 4):
                assert np.array_equal(m.smooth("GaussSeidel", psi, src, k), S2.smooth("GaussSeidel", psi, src, k)), (p["nCells"], k, "new")
            assert ctx.fallback_count() == 0
            m.close(); a.close()
        ctx.close()
    finally:
        for k in KEYS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.parametrize("direct", ["1", "0"])
def test_cluster_layout_follows_coefficient_changes(oracle, direct):
    """The cluster engine keeps its own copy of a matrix's coefficients (filled straight from the face arrays, or through
    the level layout with LDU_CLUSTER_DIRECT=0): new values, symmetric -> asymmetric -> symmetric on the SAME matrix
    object must reach it every time (GaussSeidel / DIC / DILU sweeps bit-exact after every change)."""
    saved = {k: os.environ.pop(k, None) for k in KEYS}
    os.environ.update({"LDU_CLUSTER": "2", "LDU_CLUSTER_MIN": "1", "LDU_CLUSTER_DIRECT": direct})
    try:
        ctx = capi.Context(0)
        rng = np.random.RandomState(8)
        p = cases.box3d(20, 22, 24)
        a, m = capi.from_problem(ctx, p)
        assert a.sweep_engine(2) == "clusters"
        n, nF = p["nCells"], p["lowerAddr"].size
        psi, src = rng.randn(n), rng.randn(n)
        q = dict(p)
        for step in range(6):
            up = -(0.5 + rng.rand(nF))
            q = dict(p, upper=up, diag=6.5 + rng.rand(n))
            q.pop("lower", None)
            if step % 3 == 1:
                q["lower"] = -(0.5 + rng.rand(nF))
            m.set_coeffs(q["diag"], q["upper"], q.get("lower"))
            S = oracle.System(q)
            for k in (1, 2, 4):
                assert np.array_equal(m.smooth("GaussSeidel", psi, src, k), S.smooth("GaussSeidel", psi, src, k)), (step, k)
            if "lower" in q:
                assert np.array_equal(m.precondition("DILU", src), S.precondition("DILU", src)[0]), step
                assert np.array_equal(m.precondition("DILU", src, transpose=True), S.precondition("DILU", src, transpose=True)[0]), step
            else:
                assert np.array_equal(m.precondition("DIC", src), S.precondition("DIC", src)[0]), step
            assert np.array_equal(m.Amul(psi), S.Amul(psi)), step
        assert ctx.fallback_count() == 0
        m.close(); a.close(); ctx.close()
    finally:
        for k in KEYS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.gpu
def test_split_division_is_the_compilers_division():
    """GaussSeidelSmoother.C:154 `curPsi /= diagPtr[cellI]`: the sweep kernels do the denominator's half of the IEEE
    division before the dependency wait (ldu_div, csrc/ldu_internal.hpp).  2^28 operand pairs per seed - random bit
    patterns, exponents at the edges of the fast range, zeros / denormals / huge numerators - must give the bits of the
    compiler's division."""
    ctx = capi.Context(0)
    for seed in (1, 20260928, 0xdeadbeef):
        assert ctx.div_check(1 << 28, seed) == 0


@pytest.mark.gpu
def test_cluster_trace_timeline():
    """ldu_debug_cluster_trace: every (sweep, cluster) of the pipelined cluster sweeps leaves an ordered timeline
    (start <= upper values <= lower values <= steps done <= stores acknowledged) and the traced result is unchanged."""
    import ctypes as C
    p = cases.box3d(64)
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, p)
    if a.sweep_engine(2) != "clusters":
        pytest.skip("cluster engine not selected for this size")
    L = capi.lib()
    ref = m.smooth("GaussSeidel", np.zeros(p["nCells"]), p["source"], 2)
    lev = np.zeros(4096, dtype=np.int32)
    L.ldu_debug_cluster_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    capi._chk(L.ldu_debug_cluster_levels(m.h, lev.ctypes.data, lev.size))
    nCl, nLev = int(lev[0]), int(lev[1])
    assert nCl >= p["nCells"] // 64 and 1 <= nLev < a.info()["nLevels"]
    assert lev[2] == 0 and lev[2 + nLev] == nCl and np.all(np.diff(lev[2:3 + nLev]) > 0)
    # device buffer from the HIP runtime the library itself uses (no torch here: INTEGRATION.md section 8)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    nbytes = 2 * nCl * 64
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), nbytes) == 0
    try:
        assert hip.hipMemset(buf, 0, nbytes) == 0
        assert hip.hipDeviceSynchronize() == 0
        L.ldu_debug_cluster_trace.argtypes = [C.c_void_p, C.c_void_p]
        capi._chk(L.ldu_debug_cluster_trace(m.h, buf))
        try:
            got = m.smooth("GaussSeidel", np.zeros(p["nCells"]), p["source"], 2)
        finally:
            capi._chk(L.ldu_debug_cluster_trace(m.h, None))
        T = np.zeros(2 * nCl * 8, dtype=np.int64)
        assert hip.hipMemcpy(T.ctypes.data, buf, nbytes, 2) == 0   # hipMemcpyDeviceToHost
    finally:
        hip.hipFree(buf)
    assert np.array_equal(got, ref)
    T = T.reshape(2, nCl, 8)
    assert np.all(T[:, :, 0] > 0)
    for q in range(4):
        assert np.all(T[:, :, q] <= T[:, :, q + 1])
    assert np.all(T[:, :, 5] >= 0) and np.all(T[:, :, 6] < 8)


def test_pipelined_sweeps_on_level_following_numbering_bitexact(oracle):
    """An irregular graph renumbered by Foam::bandCompression: cell indices follow the dependency levels, the XCD slabs
    lie one behind the other along the levels and every level sits in ONE slab.  1 ... 4 pipelined GaussSeidel sweeps
    (the slab engine takes all of them there), DIC and two repeats must reproduce the oracle bit for bit; the timeline
    of the level engines (ldu_debug_gs_multi_trace) must be complete and ordered."""
    import ctypes as C
    p = cases.irregular_box(56)
    order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
    nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
    p = cases.renumbered(p, order, fmap, flip, nl, nu)
    rng = np.random.RandomState(5)
    psi, src = rng.randn(p["nCells"]), rng.randn(p["nCells"])
    S = oracle.System(p)
    exp = {k: S.smooth("GaussSeidel", psi, src, k) for k in (1, 2, 3, 4)}
    dic = S.precondition("DIC", src)[0]
    for env in ({}, {"LDU_P2P_SLABS": "0"}):
        saved = {k: os.environ.pop(k, None) for k in KEYS}
        os.environ.update(env)
        os.environ["LDU_BLK"] = "0"    # (the level engines are the subject here; the block engine: test_block_engine_bitexact)
        try:
            ctx = capi.Context(0)
            a, m = capi.from_problem(ctx, p)
            for rep in range(2):
                for k in (1, 2, 3, 4):
                    assert np.array_equal(m.smooth("GaussSeidel", psi, src, k), exp[k]), (env, k, rep)
                assert np.array_equal(m.precondition("DIC", src), dic), (env, "dic", rep)
            assert ctx.fallback_count() == 0
            if not env:
                # timeline of 3 pipelined sweeps
                L = capi.lib()
                nS = a.info()["nSlices"]
                hip = C.CDLL("libamdhip64.so")
                hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
                hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
                hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                hip.hipFree.argtypes = [C.c_void_p]
                nbytes = 3 * nS * 64
                buf = C.c_void_p()
                assert hip.hipMalloc(C.byref(buf), nbytes) == 0
                try:
                    assert hip.hipMemset(buf, 0, nbytes) == 0 and hip.hipDeviceSynchronize() == 0
                    L.ldu_debug_gs_multi_trace.argtypes = [C.c_void_p, C.c_void_p]
                    capi._chk(L.ldu_debug_gs_multi_trace(m.h, buf))
                    try:
                        got = m.smooth("GaussSeidel", psi, src, 3)
                    finally:
                        capi._chk(L.ldu_debug_gs_multi_trace(m.h, None))
                    T = np.zeros(3 * nS * 8, dtype=np.int64)
                    assert hip.hipMemcpy(T.ctypes.data, buf, nbytes, 2) == 0
                finally:
                    hip.hipFree(buf)
                assert np.array_equal(got, exp[3])
                T = T.reshape(3, nS, 8)
                assert np.all(T[:, :, 0] > 0)
                for q in range(3):
                    assert np.all(T[:, :, q] <= T[:, :, q + 1])
        finally:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update({k: v for k, v in saved.items() if v is not None})
