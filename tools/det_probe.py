"""Determinism / parity probe at a given box size: every engine config against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as entry

entry.load_package()
from openfoam_amd import capi, cases
import oracle_py
import torch

nx, ny, nz = [int(x) for x in sys.argv[1:4]]
p = cases.box3d(nx, ny, nz)
nC = p["nCells"]
S = oracle_py.System([p])
w_ref, _ = S.precondition("DIC", p["source"])
ps_ref = S.smooth("GaussSeidel", np.zeros(nC), p["source"], 3)
dev = torch.device("cuda", 0)
d_src = torch.from_numpy(p["source"]).to(dev)
import json
CFG = json.loads(os.environ.get("PROBE_CFGS", "{}")) or {"chip": {"LDU_P2P_SLABS": "0"}, "auto": {}}
for name, env in CFG.items():
    for k in ("LDU_P2P_BPC", "LDU_P2P_SLABS", "LDU_P2P_PROXY", "LDU_P2P_SLEEP", "LDU_P2P_MAXBPC", "LDU_CLUSTER",
              "LDU_CLUSTER_MIN", "LDU_CLUSTER_BPC", "LDU_CLUSTER_BPC_MULTI"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, p)
    L = capi.lib()
    d_w = torch.zeros(nC, dtype=torch.float64, device=dev)
    for rep in range(2):
        d_psi = torch.zeros(nC, dtype=torch.float64, device=dev)
        capi._chk(L.ldu_precondition(m.h, 2, capi._ptr(d_w), capi._ptr(d_src), 0))
        torch.cuda.synchronize()
        m.profile_begin()
        capi._chk(L.ldu_precondition(m.h, 2, capi._ptr(d_w), capi._ptr(d_src), 0))
        capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), 1))
        capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), 2))
        prof = m.profile_end()
        w = d_w.cpu().numpy(); ps = d_psi.cpu().numpy()
        bw = np.flatnonzero(w != w_ref); bp = np.flatnonzero(ps != ps_ref)
        t = {k: v["ms"] / v["count"] for k, v in prof.items()}
        print("%-15s rep %d: DIC diff %d GS diff %d | tri %.3f gs %.3f gs_multi(2) %.3f ms" % (
            name, rep, bw.size, bp.size, t.get("tri_sweep", 0), t.get("gs_sweep", 0), t.get("gs_multi", 0)), flush=True)
    m.close(); a.close(); ctx.close()
