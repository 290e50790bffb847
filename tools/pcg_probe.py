"""PCG+DIC iterations/s on the symmetric n^3 box (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
p = cases.box3d(n)
dev = torch.device("cuda", 0)
d_src = torch.from_numpy(p["source"]).to(dev)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
d_psi = torch.zeros(p["nCells"], dtype=torch.float64, device=dev)
kw = dict(history=False, solver="PCG", preconditioner="DIC", tolerance=0.0, relTol=0.0)
m.solve(d_psi, d_src, maxIter=4, **kw)
d_psi.zero_(); torch.cuda.synchronize()
t0 = time.perf_counter()
_, perf = m.solve(d_psi, d_src, maxIter=iters, **kw)
dt = time.perf_counter() - t0
print("PCG+DIC %d iterations in %.1f ms -> %.1f it/s  final residual %.3e" % (perf["nIterations"], dt * 1e3, perf["nIterations"] / dt, perf["finalResidual"]), flush=True)
m.close(); a.close(); ctx.close()
