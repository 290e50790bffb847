"""Replay of fuzz case 8723 (seed 31337): three-plane DILU on random_graph(52469, 3, 5, asym=True)."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import conftest  # noqa
import oracle_py as O
from openfoam_amd import capi, cases
p = cases.random_graph(52469, 3, 5, asym=True)
rng = np.random.RandomState(1)
n = p["nCells"]
S3 = rng.randn(n, 3); P3 = rng.randn(n, 3)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
S = O.System(p)
print("info", a.info(), flush=True)
for name, g, o in [("DILU", lambda: m.precondition("DILU", S3[:, 0].copy()), lambda: S.precondition("DILU", S3[:, 0].copy())[0]),
                   ("cDILU", lambda: m.coupled_precondition("DILU", S3), lambda: S.c_precondition("DILU", S3)),
                   ("cGS", lambda: m.coupled_smooth(P3, S3, 2), lambda: S.c_smooth(P3, S3, 2))]:
    for rep in range(3):
        try:
            got = g()
            print(name, rep, "equal" if np.array_equal(got, o()) else "MISMATCH", flush=True)
        except Exception as e:
            print(name, rep, "ERROR", e, flush=True)
