"""Small fixed workload for PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): on the 216^3 box
run 3x {dot product (known bytes: calibration), Amul, 2 pipelined GaussSeidel sweeps, DIC apply}."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
from openfoam_amd import capi, cases
import torch
import ctypes as C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
p = cases.box3d(n)
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
nC = p["nCells"]
d_src = torch.from_numpy(p["source"]).to(dev)
d_x = torch.zeros(nC, dtype=torch.float64, device=dev)
d_y = torch.zeros(nC, dtype=torch.float64, device=dev)
L = capi.lib()
r = C.c_double()
for _ in range(3):
    capi._chk(L.ldu_gSumProd(m.h, capi._ptr(d_src), capi._ptr(d_src), C.byref(r)))
    capi._chk(L.ldu_amul(m.h, capi._ptr(d_y), capi._ptr(d_src)))
    capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_x), capi._ptr(d_src), 2))
    capi._chk(L.ldu_precondition(m.h, 2, capi._ptr(d_y), capi._ptr(d_src), 0))
print("done", r.value)
m.close(); a.close(); ctx.close()
