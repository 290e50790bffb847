"""EXPERIMENT: where does a cluster spend its time?  (t0 start, t1 inputs loaded, t2 externals arrived, t3 done)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
import torch, ctypes as C
n = int(sys.argv[1])
os.environ["LDU_CLUSTER"] = "2"; os.environ["LDU_CLUSTER_MIN"] = "1000"
p = cases.box3d(n)
ctx = capi.Context(0); a, m = capi.from_problem(ctx, p)
L = capi.lib()
dev = torch.device("cuda", 0)
src = torch.from_numpy(p["source"]).to(dev); w = torch.zeros_like(src); psi = torch.zeros_like(src)
torch.cuda.synchronize()
nCl = (p["nCells"] + 63) // 64 + 1000
buf = torch.zeros(nCl * 4, dtype=torch.int64, device=dev)
for what in ("DIC", "GS"):
    for rep in range(2):
        if what == "DIC": capi._chk(L.ldu_precondition(m.h, 2, capi._ptr(w), capi._ptr(src), 0))
        else: capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(psi), capi._ptr(src), 1))
    buf.zero_(); torch.cuda.synchronize()
    L.ldu_debug_cl_trace(C.c_void_p(buf.data_ptr()))
    if what == "DIC":
        # only the forward sweep is traced last? trace both: the backward overwrites records of the forward
        capi._chk(L.ldu_precondition(m.h, 2, capi._ptr(w), capi._ptr(src), 0))
    else:
        capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(psi), capi._ptr(src), 1))
    L.ldu_debug_cl_trace(C.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
    t = t[t[:, 3] > 0]
    f = 0.01  # wall_clock64: 100 MHz -> 10 ns ticks -> us
    load, wait, steps, total = (t[:, 1] - t[:, 0]) * f, (t[:, 2] - t[:, 1]) * f, (t[:, 3] - t[:, 2]) * f, (t[:, 3] - t[:, 0]) * f
    span = (t[:, 3].max() - t[:, 0].min()) * f
    print("%s n=%d clusters=%d kernel span %.1f us | per cluster: load %.2f  wait-for-externals %.2f  steps %.2f  total %.2f us (medians: %.2f %.2f %.2f)"
          % (what, n, len(t), span, load.mean(), wait.mean(), steps.mean(), total.mean(), np.median(load), np.median(wait), np.median(steps)))
