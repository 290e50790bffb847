"""Timeline of k pipelined GaussSeidel sweeps on the level engines (ldu_debug_gs_multi_trace) on the irregular,
bandCompression-renumbered stand-in: where is every sweep's front at which time?
python tools/gsm_trace.py [n=216] [k=4]"""
import os, sys, ctypes as C, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch
torch.cuda.init()
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = cases.irregular_box(n)
order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
p = cases.renumbered(p, order, fmap, flip, nl, nu)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
info = a.info()
L = capi.lib()
rng = np.random.RandomState(1)
dx, db = torch.from_numpy(rng.randn(p["nCells"])).cuda(), torch.from_numpy(rng.randn(p["nCells"])).cuda()
def sweep():
    capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(dx), capi._ptr(db), k)); ctx.sync()
sweep(); sweep()
lev = np.zeros(info["nLevels"] + 8, dtype=np.int32)
L.ldu_debug_slice_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
capi._chk(L.ldu_debug_slice_levels(m.h, lev.ctypes.data, lev.size))
nLev = int(lev[0]); start = lev[1:2 + nLev]; nS = int(start[-1])
buf = torch.zeros(k * nS * 8, dtype=torch.int64, device="cuda")
L.ldu_debug_gs_multi_trace.argtypes = [C.c_void_p, C.c_void_p]
capi._chk(L.ldu_debug_gs_multi_trace(m.h, C.c_void_p(buf.data_ptr())))
t0 = time.perf_counter(); sweep(); dt = time.perf_counter() - t0
capi._chk(L.ldu_debug_gs_multi_trace(m.h, None))
T = buf.cpu().numpy().reshape(k, nS, 8).astype(np.float64)
tmin = T[:, :, 0][T[:, :, 0] > 0].min()
T[:, :, :4] = (T[:, :, :4] - tmin) * 0.01
print("irregular %d^3: %d levels, %d slices, engine %s, %d sweeps traced in %.2f ms" % (n, nLev, nS, a.sweep_engine(2), k, dt * 1e3))
for j in range(k):
    X = T[j]
    done = np.array([X[start[l]:start[l + 1], 3].max() for l in range(nLev)])
    first = np.array([X[start[l]:start[l + 1], 0].min() for l in range(nLev)])
    d = np.diff(done)
    print("sweep %d: level 0 done at %.0f us, last level at %.0f us; per level mean %.2f us p50 %.2f p90 %.2f p99 %.2f max %.1f"
          % (j, done[0], done[-1], d.mean(), *np.percentile(d, [50, 90, 99]), d.max()))
    print("   per task (medians): start->upper %.2f  upper->lower(ready) %.2f  ready->stored %.2f us; tasks start %.1f us before their level completes"
          % (np.median(X[:, 1] - X[:, 0]), np.median(X[:, 2] - X[:, 1]), np.median(X[:, 3] - X[:, 2]),
             np.median(np.repeat(done, np.diff(start)) - X[:, 0])))
    q = max(1, nLev // 12)
    print("   level: done-time [us] XCC of its tasks:", "  ".join("%d: %.0f x%s" % (l, done[l], "".join(str(int(v)) for v in sorted(set(X[start[l]:start[l + 1], 4]))))
                                                            for l in range(0, nLev, q)))
