"""BASELINE config C5 - "interFoam damBreak ..., 2-phase PISO with GAMG p_rgh" - through an UNCHANGED reference application
(VERDICT r3 item 8): oracle/_ref/interFoam is the reference's own applications/solvers/multiphase/interFoam/interFoam.C,
compiled where it lies with the units of its transport / interface / turbulence libraries (oracle/build_ref_interfoam.sh, no
stand-ins); the case is the tutorial's (oracle/dambreak_case.py), on the mesh the reference's blockMesh made and the phase
fraction the reference's setFields wrote (tests/golden/damBreak_2268.npz, tests/golden/make_dambreak_golden.py).
 * CPU: the stock run to t = 0.1 reproduces the committed log fixtures (tests/golden/interfoam_dambreak*.json).
 * GPU (-m gpu): the same binary and case plus `libs ("libhipLduSolvers.so");`: every pcorr / p_rgh solve (PCG + DIC with
   the density-jump coefficients of the two-phase p_rgh equation, or GAMG as BASELINE words C5) on the GPU library, held to the
   stock run's lines (tolerance below).  The ~4 M-cell size of C5 is covered by the 2-D jump twin (tests/test_gpu_fullsize.py)."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import cavity_case as cc
import dambreak_case as dc

PLUGIN = os.path.abspath(os.path.join(HERE, "..", "openfoam-2.2.x_amd", "lib", "libhipLduSolvers.so"))
END = 0.1
needs_ref = pytest.mark.skipif(not dc.available(), reason="needs oracle/_ref/interFoam (oracle/build_ref_interfoam.sh)")
CASES = [("", None), ("_gamg", dc.GAMG)]


def golden(tag):
    return [tuple(l) for l in json.load(open(os.path.join(HERE, "golden", "interfoam_dambreak%s.json" % tag)))["lines"]]


@needs_ref
@pytest.mark.parametrize("tag,psolver", CASES, ids=["pcg", "gamg"])
def test_stock_interfoam_reproduces_the_fixture(tag, psolver, tmp_path):
    case = str(tmp_path / "damBreak")
    dc.write(case, END, p_solver=psolver)
    assert cc.solve_lines(dc.run(case)) == golden(tag)


@pytest.mark.gpu
@needs_ref
@pytest.mark.skipif(not os.path.exists(PLUGIN), reason="needs the prebuilt plugin")
@pytest.mark.parametrize("tag,psolver", CASES, ids=["pcg", "gamg"])
def test_interfoam_through_the_plugin(tag, psolver, tmp_path):
    case = str(tmp_path / "damBreak")
    dc.write(case, END, libs=[PLUGIN], p_solver=psolver)
    log = dc.run(case, extra_env={"LDU_VERBOSE": "1"})
    assert "[hipLduSolvers]" in log, log[-2000:]
    lines, gold = cc.solve_lines(log), golden(tag)
    # the time step is adjusted from the Courant number of the computed flow: the same number of solver lines means the same
    # sequence of time steps
    assert len(lines) == len(gold)
    # Every solve stops at a relative tolerance and feeds the next (3 PISO correctors per step, the last to relTol 0): the
    # comparison is by solver name, field, iteration count (within 2 on at most 5 % of the lines: a residual landing on the
    # threshold) and initial residual to 1e-3 + the noise floor of a converged corrector (1e-3 x the 1e-7 tolerance)
    off, worst = 0, 0.0
    for got, ref in zip(lines, gold):
        assert got[0] == ref[0] and got[1] == ref[1], (got, ref)
        assert abs(got[4] - ref[4]) <= 2, (got, ref)
        off += int(got[4] != ref[4])
        assert abs(got[2] - ref[2]) <= 1e-3 * abs(ref[2]) + 1e-10, (got, ref)
        if ref[2]:
            worst = max(worst, abs(got[2] - ref[2]) / abs(ref[2]))
    print("interFoam damBreak through the plugin (%s): %d solver lines, %d with a different iteration count, worst relative "
          "difference of an initial residual %.2e" % (tag or "pcg", len(lines), off, worst))
    assert off <= len(lines) // 20
