import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
import oracle_py, numpy as np
p = cases.random_graph(200000, avg_deg=9, band=3000, seed=3)
S = oracle_py.System(p)
xo = S.smooth("GaussSeidel", p["psi"], p["source"], 2)
for env in ({"LDU_P2P_MAXBPC": "1"}, {"LDU_P2P_MAXBPC": "2"}, {"LDU_P2P_MAXBPC": "3"}, {"LDU_P2P_MAXBPC": "5"}, {"LDU_P2P_MAXBPC": "5", "LDU_GS_FAST": "1"}):
    os.environ.update(env)
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, p)
    res = []
    for rep in range(3):
        x = m.smooth("GaussSeidel", p["psi"], p["source"], 2)
        ok = np.array_equal(x, xo)
        try:
            m.gSumMag(x); err = ""
        except Exception as e:
            err = "ABORT"
        res.append((ok, err))
    print(env, res, flush=True)
    m.close(); a.close(); ctx.close()
