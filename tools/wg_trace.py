"""Timeline of the one-workgroup GaussSeidel engine (gs_wg_kernel, ldu_debug_gs_multi_trace) on the small GAMG levels of the
216^3 hierarchy (the same level matrices: the levels below 157 464 cells of a 54^3 box are the levels below 157 464 cells of
the 216^3 box): per (sweep, slice) task the 100 MHz clock at step start / loads issued / dependencies seen / value stored.
python tools/wg_trace.py [k=4]"""
import os, sys, ctypes as C, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import torch
torch.cuda.init()
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
import oracle_py as O
O.build()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = cases.box3d(54)
S = O.System([p])
levels = S.gamg_levels(smoother="GaussSeidel", nCellsInCoarsestLevel=10, mergeLevels=1, agglomerator="faceAreaPair")
ctx = capi.Context(0)
L = capi.lib()
L.ldu_debug_slice_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
L.ldu_debug_gs_multi_trace.argtypes = [C.c_void_p, C.c_void_p]
rng = np.random.RandomState(1)
for lv in levels:
    n = lv["nCells"]
    if n > 5000 or n < 30:
        continue
    q = dict(nCells=n, lowerAddr=lv["lowerAddr"], upperAddr=lv["upperAddr"], diag=lv["diag"], upper=lv["upper"])
    a = capi.Addressing(ctx, n, q["lowerAddr"], q["upperAddr"])
    m = capi.Matrix(a)
    m.set_coeffs(q["diag"], q["upper"], None)
    dx, db = torch.from_numpy(rng.randn(n)).cuda(), torch.from_numpy(rng.randn(n)).cuda()

    def sweep():
        capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(dx), capi._ptr(db), k)); ctx.sync()
    sweep(); sweep()
    t0 = time.perf_counter()
    for _ in range(20):
        capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(dx), capi._ptr(db), k))
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    info = a.info()
    lev = np.zeros(info["nLevels"] + 8, dtype=np.int32)
    capi._chk(L.ldu_debug_slice_levels(m.h, lev.ctypes.data, lev.size))
    nLev = int(lev[0]); nS = int(lev[1:2 + nLev][-1])
    nT = k * nS
    buf = torch.zeros(nT * 8 + 64, dtype=torch.int64, device="cuda")
    capi._chk(L.ldu_debug_gs_multi_trace(m.h, C.c_void_p(buf.data_ptr())))
    sweep()
    capi._chk(L.ldu_debug_gs_multi_trace(m.h, None))
    T = buf.cpu().numpy()[:nT * 8].reshape(nT, 8).astype(np.float64)
    ok = T[:, 3] > 0
    T = T[ok]
    tmin = T[:, 0].min()
    T[:, :4] = (T[:, :4] - tmin) * 0.01   # us
    span = T[:, 3].max()
    order = np.argsort(T[:, 3])
    gaps = np.diff(T[order, 3])
    print("level of %5d cells, engine %s: %d levels, %d slices, %d tasks traced; %d sweeps %.1f us per call (untraced), traced "
          "kernel span %.1f us = %.2f us per task" % (n, a.sweep_engine(2), nLev, nS, T.shape[0], k, dt * 1e6, span, span / max(1, T.shape[0])))
    print("   per task, medians [us]: start -> loads issued %.2f, loads issued -> dependencies seen %.2f, seen -> stored %.2f; "
          "p90 %.2f / %.2f / %.2f" % (np.median(T[:, 1] - T[:, 0]), np.median(T[:, 2] - T[:, 1]), np.median(T[:, 3] - T[:, 2]),
                                      *np.percentile(T[:, 1] - T[:, 0], [90]), *np.percentile(T[:, 2] - T[:, 1], [90]),
                                      *np.percentile(T[:, 3] - T[:, 2], [90])))
    print("   first task starts at %.1f us, first value stored at %.1f us; time between consecutive stores: median %.2f p90 %.2f us; "
          "tasks per wavefront %d" % (T[:, 0].min(), T[:, 3].min(), np.median(gaps), np.percentile(gaps, 90),
                                      int(np.max(np.bincount(T[:, 4].astype(int))))))
    m.close(); a.close()
