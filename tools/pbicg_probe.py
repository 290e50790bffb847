"""PBiCG+DILU iterations/s on the asymmetric n^3 box, dual-stream on/off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
p = cases.box3d(n, asym=True)
dev = torch.device("cuda", 0)
d_src = torch.from_numpy(p["source"]).to(dev)
for dual in ("1", "0"):
    os.environ["LDU_DUAL_STREAM"] = dual
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, p)
    d_psi = torch.zeros(p["nCells"], dtype=torch.float64, device=dev)
    m.solve(d_psi, d_src, history=False, solver="PBiCG", preconditioner="DILU", tolerance=0.0, relTol=0.0, maxIter=4)
    d_psi.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, perf = m.solve(d_psi, d_src, history=False, solver="PBiCG", preconditioner="DILU", tolerance=0.0, relTol=0.0, maxIter=29)
    dt = time.perf_counter() - t0
    nC, nF = p["nCells"], p["lowerAddr"].size
    print("dual_stream=%s  PBiCG+DILU %d iterations in %.1f ms -> %.1f it/s (%.0f GB/s algorithmic 256nC+120nF)  final residual %.3e"
          % (dual, perf["nIterations"], dt * 1e3, perf["nIterations"] / dt, (256.0 * nC + 120.0 * nF) * perf["nIterations"] / dt / 1e9, perf["finalResidual"]), flush=True)
    m.close(); a.close(); ctx.close()
