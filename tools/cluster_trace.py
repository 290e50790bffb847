"""Per-task timeline of the pipelined GaussSeidel cluster sweeps (ldu_debug_cluster_trace).
python tools/cluster_trace.py [n=216] [sweeps=2]

Prints, per cluster level of sweep 0 (every 10th), when its tasks became ready / finished, and the split of a
level's time into: steps + stores of the producers, and hand-off (producer's store acknowledged -> consumer's poll
succeeds)."""
import os, sys, ctypes as C
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import __graft_entry__ as entry
entry.load_package()
import torch
from openfoam_amd import capi, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = cases.box3d(n)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
dev = torch.device("cuda", 0)
L = capi.lib()
d_src = torch.from_numpy(p["source"]).to(dev)
d_psi = torch.zeros(p["nCells"], dtype=torch.float64, device=dev)
def sweep():
    d_psi.zero_(); torch.cuda.synchronize()
    capi._chk(L.ldu_smooth(m.h, capi.SMOOTHERS["GaussSeidel"], capi._ptr(d_psi), capi._ptr(d_src), k))
    torch.cuda.synchronize()
sweep(); sweep()
lev = np.zeros(4096, dtype=np.int32)
L.ldu_debug_cluster_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
capi._chk(L.ldu_debug_cluster_levels(m.h, lev.ctypes.data, lev.size))
nCl, nLev = int(lev[0]), int(lev[1]); start = lev[2:3 + nLev]
buf = torch.zeros(k * nCl * 8, dtype=torch.int64, device=dev)
L.ldu_debug_cluster_trace.argtypes = [C.c_void_p, C.c_void_p]
capi._chk(L.ldu_debug_cluster_trace(m.h, C.c_void_p(buf.data_ptr())))
import time
t0 = time.perf_counter(); sweep(); dt = time.perf_counter() - t0
capi._chk(L.ldu_debug_cluster_trace(m.h, None))
T = buf.cpu().numpy().reshape(k, nCl, 8).astype(np.float64)
tmin = T[:, :, 0][T[:, :, 0] > 0].min()
for j in range(k):
    T[j, :, :5] = (T[j, :, :5] - tmin) * 0.01     # us (100 MHz)
print("%d^3: %d clusters, %d cluster levels, %d sweeps traced: %.3f ms wall (tracing on)" % (n, nCl, nLev, k, dt * 1e3))
for j in range(k):
    X = T[j]
    print("sweep %d: first start %.1f us, last stored %.1f us" % (j, X[:, 0].min(), X[:, 4].max()))
    print("  per task: start->upper %.2f  upper->lower(ready) %.2f  ready->steps done %.2f  steps->stored %.2f us (medians); polls median %.0f mean %.1f"
          % tuple([np.median(X[:, 1] - X[:, 0]), np.median(X[:, 2] - X[:, 1]), np.median(X[:, 3] - X[:, 2]), np.median(X[:, 4] - X[:, 3]),
                   np.median(X[:, 5]), X[:, 5].mean()]))
    # front: per level the time the LAST task became ready / stored, the FIRST became ready
    rl = np.array([X[start[l]:start[l + 1], 2].max() for l in range(nLev)])
    rf = np.array([X[start[l]:start[l + 1], 2].min() for l in range(nLev)])
    rm = np.array([np.median(X[start[l]:start[l + 1], 2]) for l in range(nLev)])
    sm = np.array([np.median(X[start[l]:start[l + 1], 4]) for l in range(nLev)])
    sd = np.array([np.median(X[start[l]:start[l + 1], 3]) for l in range(nLev)])
    st0 = np.array([np.median(X[start[l]:start[l + 1], 0]) for l in range(nLev)])
    d = np.diff(rm)
    print("  median-ready front: %.2f us per cluster level (mean over levels), p10 %.2f p50 %.2f p90 %.2f" % (d.mean(), *np.percentile(d, [10, 50, 90])))
    print("  of a level's %.2f us: ready->steps done %.2f, steps done->stores acknowledged %.2f, stored(L)->ready(L+1) %.2f"
          % (d.mean(), np.mean(sd - rm), np.mean(sm - sd), np.mean(rm[1:] - sm[:-1])))
    print("  run-ahead: tasks start %.1f us (median) before they become ready" % np.median(X[:, 2] - X[:, 0]))
    print("  level   tasks  start(med)  ready(first med last)  stored(med)")
    for l in range(0, nLev, max(1, nLev // 16)):
        print("  %5d  %6d  %9.1f   %8.1f %8.1f %8.1f   %8.1f" % (l, start[l + 1] - start[l], st0[l], rf[l], rm[l], rl[l], sm[l]))
