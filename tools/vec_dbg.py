import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import conftest  # noqa
import oracle_py as O
from openfoam_amd import capi, cases
os.environ["LDU_CLUSTER"] = "2"; os.environ["LDU_CLUSTER_MIN"] = "1"; os.environ["LDU_VERBOSE"] = "1"
probs = {"box_asym": cases.box3d(17, 30, 21, asym=True), "graph_sparse": cases.random_graph(20000, 2, 60, asym=True),
         "box_small": cases.box3d(5, 4, 3, asym=True)}
ctx = capi.Context(0)
for name, p in probs.items():
    S = O.System(p)
    rng = np.random.RandomState(21)
    psi, src = rng.randn(p["nCells"], 3), rng.randn(p["nCells"], 3)
    a, m = capi.from_problem(ctx, p)
    for what in ("pre", "preT", "gs"):
        try:
            if what == "pre": g, e = m.coupled_precondition("DILU", src), S.c_precondition("DILU", src)
            elif what == "preT": g, e = m.coupled_precondition("DILU", src, True), S.c_precondition("DILU", src, True)
            else: g, e = m.coupled_smooth(psi, src, 2), S.c_smooth(psi, src, 2)
            print(name, what, "equal", np.array_equal(g, e), "maxdiff", np.abs(g - e).max(), flush=True)
        except Exception as ex:
            print(name, what, "EXC", str(ex)[:100], flush=True)
    m.close(); a.close()
