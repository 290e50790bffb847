"""CPU: the degenerate inputs of test_gpu_edge_cases.py through the REFERENCE (oracle/_ref) and the oracle."""
import numpy as np
import pytest

import oracle_py
from test_gpu_edge_cases import CASES

pytestmark = pytest.mark.skipif(not oracle_py.ref_available(), reason="needs oracle/_ref (build_ref.sh)")

SOLVERS = [dict(solver="PCG", preconditioner="DIC"), dict(solver="PBiCG", preconditioner="DILU"),
           dict(solver="smoothSolver", smoother="GaussSeidel", maxIter=60)]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_on_degenerate_systems(oracle, name):
    p = dict(CASES[name])
    S = oracle.System(p)
    for kw in SOLVERS:
        if kw["solver"] == "PCG" and "lower" in p:
            continue
        kw = dict(kw, tolerance=1e-10, relTol=0)
        q = dict(p, lower=p["upper"].copy()) if kw["solver"] == "PBiCG" and "lower" not in p else p
        ref, _ = oracle_py.run_ref("solve", q, oracle_py.dict_string(**kw))
        okw = dict(kw)
        if "preconditioner" in okw:
            okw["precond"] = okw.pop("preconditioner")
        x, perf = oracle.System(q).solve(q["psi"], q["source"], **okw)
        assert perf["nIterations"] == int(ref["perf"][2]), (name, kw)
        assert np.array_equal([perf["initialResidual"], perf["finalResidual"]], ref["perf"][:2]), (name, kw)
        assert np.array_equal(x, ref["psi"]), (name, kw)
