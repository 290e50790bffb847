"""The C oracle against the committed golden vectors (outputs of the real reference,
tests/golden/make_golden.py).  Runs anywhere (no /root/reference needed)."""
import os
import sys

import numpy as np
import pytest

from openfoam_amd import ldub

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402


def _okw(kw):
    kw = dict(kw)
    kw.pop("cacheAgglomeration", None)
    return kw


@pytest.mark.parametrize("name", sorted(make_golden.PROBLEMS))
def test_oracle_matches_golden(name, oracle):
    gen, solves = make_golden.PROBLEMS[name]
    p = gen()
    G = ldub.read(os.path.join(HERE, "golden", name + ".ldub"))
    S = oracle.System(p)
    assert np.array_equal(S.Amul(p["psi"]), G["ops_Amul"])
    assert np.array_equal(S.residual(p["psi"], p["source"]), G["ops_residual"])
    assert np.array_equal(S.sumA(), G["ops_sumA"])
    assert np.array_equal(S.smooth("GaussSeidel", p["psi"], p["source"], 1), G["ops_smooth1_GaussSeidel"])
    if S.sym:
        assert np.array_equal(S.precondition("DIC", p["source"])[0], G["ops_precond_DIC"])
    else:
        assert np.array_equal(S.precondition("DILU", p["source"])[0], G["ops_precond_DILU"])
        assert np.array_equal(S.precondition("DILU", p["source"], transpose=True)[0], G["ops_precondT_DILU"])
    for i, (sname, kw) in enumerate(solves):
        x, perf = S.solve(p["psi"], p["source"], **_okw(kw))
        gp = G["solve%d_perf" % i]
        assert perf["nIterations"] == int(gp[2]), (name, sname)
        assert perf["initialResidual"] == gp[0]
        assert perf["finalResidual"] == gp[1]
        assert np.array_equal(x, G["solve%d_psi" % i])
        h = G["solve%d_hist" % i]
        n = min(len(h), len(perf["history"]))
        assert np.array_equal(h[:n], perf["history"][:n])
        if kw["solver"] == "GAMG":
            lv = S.gamg_levels(**_okw(kw))
            assert [L["nCells"] for L in lv] == list(G["solve%d_nCellsPerLevel" % i])
            assert np.array_equal(lv[0]["restrict"], G["solve%d_restrict0" % i])
            assert np.array_equal(lv[-1]["diag"], G["solve%d_coarsestDiag" % i])


def test_known_answer_survey(oracle):
    """SURVEY.md 8c / BASELINE.md 2: DICPCG on the 40x40 Laplacian: 78 iterations, 9.4088e-11."""
    from openfoam_amd import cases
    p = cases.laplacian2d(40, 40)
    x, perf = oracle.System(p).solve(p["psi"], p["source"], solver="PCG", preconditioner="DIC",
                                     tolerance=1e-10, relTol=0)
    assert perf["nIterations"] == 78
    assert abs(perf["finalResidual"] - 9.4088e-11) < 1e-15
    assert abs(x[0] - 1.67566) < 1e-5 and abs(x[1599] - 1.48314) < 1e-5
