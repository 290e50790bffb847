"""BASELINE config C1 - "icoFoam cavity tutorial 40x40x1, PCG/DIC for p" - through an UNCHANGED reference application:
oracle/_ref/icoFoam is the reference's own applications/solvers/incompressible/icoFoam/icoFoam.C, compiled where it lies
and linked against the reference's libfiniteVolume units + libOpenFOAM (oracle/build_ref_fv.sh, no stand-ins).
 * CPU: the stock run reproduces the committed log fixture (pins tests/golden/icofoam_cavity40*.json).
 * GPU (-m gpu): the same binary, the same case, plus `libs ("libhipLduSolvers.so");` in system/controlDict: every
   `Solving for Ux|Uy|p` line of the 100 time steps - same solver name, same `No Iterations`, residuals to the printed
   digits (tolerance below) - with the GPU library doing every solve (U: PBiCG/DILU, p: PCG/DIC; second fixture: p with
   the motorBike GAMG block, faceAreaPair weights read by the shim from the fvMesh behind the matrix)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import cavity_case as cc

PLUGIN = os.path.abspath(os.path.join(HERE, "..", "openfoam-2.2.x_amd", "lib", "libhipLduSolvers.so"))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_icofoam_golden import GAMG

needs_ref = pytest.mark.skipif(not cc.available(), reason="needs oracle/_ref/icoFoam (oracle/build_ref_fv.sh)")


def golden(tag):
    return [tuple(l) for l in json.load(open(os.path.join(HERE, "golden", "icofoam_cavity40%s.json" % tag)))["lines"]]


@needs_ref
@pytest.mark.parametrize("tag,psolver", [("", None), ("_gamg", GAMG)], ids=["pcg", "gamg"])
def test_stock_icofoam_reproduces_the_fixture(tag, psolver, tmp_path):
    case = str(tmp_path / "cavity")
    cc.write(case, 40, 100, p_solver=psolver)
    lines = cc.solve_lines(cc.run(case))
    assert lines == golden(tag)


def compare(lines, gold):
    """same solver, same field, same `No Iterations` on every line; residuals: 400 consecutive solves feed each other
    (every solve stops at a tolerance, so the next one starts from a field that differs in the last digits, and PCG /
    PBiCG amplify that by the time they reach 1e-6), so the bar is the solver-level one (1e-6) on the INITIAL residual of
    a time step's first solves; the printed six digits agree on most lines (reported)."""
    assert len(lines) == len(gold) == 400
    worst, same = 0.0, 0
    for got, ref in zip(lines, gold):
        assert got[0] == ref[0] and got[1] == ref[1], (got, ref)          # solver name, field
        assert got[4] == ref[4], (got, ref)                                # No Iterations
        same += int(got[2] == ref[2] and got[3] == ref[3])
        # initial residual (the state the solve starts from): 2e-5; final residual (after up to 70 Krylov iterations,
        # which amplify the summation-order differences of the global sums): 1e-3 - its role is the convergence test, and
        # `No Iterations` above is equal on every line.  One unit of the last printed digit on top: a value next to a
        # rounding boundary flips it under any difference.
        # Absolute floor 2e-10 = 2e-4 x the p tolerance (1e-6): once the flow is steady the initial residuals sit at the
        # solver tolerance and every solve stops somewhere below it, so they carry the previous solves' stopping noise.
        for a, b, rt in ((got[2], ref[2], 2e-5), (got[3], ref[3], 1e-3)):
            tol = rt * abs(b) + 2e-10 + 1.0 * 10.0 ** (np.floor(np.log10(abs(b))) - 5) if b else 1e-300
            assert abs(a - b) <= tol, (got, ref)
            if b:
                worst = max(worst, abs(a - b) / abs(b))
    return worst, same


@pytest.mark.gpu
@needs_ref
@pytest.mark.skipif(not os.path.exists(PLUGIN), reason="needs the prebuilt plugin")
@pytest.mark.parametrize("tag,psolver", [("", None), ("_gamg", GAMG)], ids=["pcg", "gamg"])
def test_icofoam_through_the_plugin(tag, psolver, tmp_path):
    case = str(tmp_path / "cavity")
    cc.write(case, 40, 100, libs=[PLUGIN], p_solver=psolver)
    log = cc.run(case, extra_env={"LDU_VERBOSE": "1"})
    # the plugin really carried the solves (it announces itself once per solver type)
    assert "[hipLduSolvers]" in log, log[-2000:]
    worst, same = compare(cc.solve_lines(log), golden(tag))
    print("icoFoam cavity 40x40 through the plugin: 400 solver lines, iteration counts equal; %d lines identical to the "
          "printed digit, worst relative residual difference %.2e" % (same, worst))
    assert same >= 300


@needs_ref
def test_icofoam_binary_is_the_unchanged_application():
    """the drop-in claim: the application carries no product symbol"""
    import subprocess
    out = subprocess.run(["nm", "-C", os.path.join(cc.REF, "icoFoam")], capture_output=True, text=True).stdout
    assert "hipLdu" not in out and "ldu_solve" not in out
