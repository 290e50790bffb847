"""Probe: GaussSeidel (1 and k pipelined sweeps), DIC half sweep and Amul on one mesh for several engine settings,
with an optional per-level timeline of the pipelined sweeps on the level engines.
usage: python tools/mesh_probe.py MESH [cfg ...]
  MESH = box:N | octree:Q:Lmin:Lmax[:hexref] | irregular:N, optionally followed by @K = the K-th GAMG coarse level of that
         matrix (faceAreaPair agglomeration by the CPU oracle - tools only), e.g. octree:14:6:7@1
  cfg  = name:ENV=VAL,ENV=VAL        (PROBE_SWEEPS=k, PROBE_TRACE=1 are read per configuration)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
from openfoam_amd import capi, cases, octree, motorbike
import torch


def make(spec):
    f = spec.split(":")
    if f[0] == "box":
        return cases.box3d(int(f[1]))
    if f[0] == "chain":            # chain:N - one row per dependency level: the bare hand-off latency of an engine
        return cases.laplacian2d(1, int(f[1]))
    if f[0] == "grid2d":           # grid2d:N - N x N five-point matrix
        return cases.laplacian2d(int(f[1]), int(f[1]))
    if f[0] == "irregular":
        p = cases.irregular_box(int(f[1]))
    elif f[0] == "motorbike":      # motorbike:<stored mesh>[:snappy]  (renumbered by Foam::bandCompression unless :snappy)
        p = motorbike.problem(f[1])
        p.pop("cellLevel"); p.pop("meta")
        if len(f) > 2 and f[2] == "snappy":
            return p
    elif f[0] == "octree":
        q = int(f[1])
        p = octree.problem(base=(5 * q, 2 * q, 2 * q), surface_levels=(int(f[2]), int(f[3])))
        p.pop("cellLevel")
        if len(f) > 4 and f[4] == "hexref":
            return p
    else:
        raise SystemExit("unknown mesh " + spec)
    order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
    nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
    return cases.renumbered(p, order, fmap, flip, nl, nu)


spec = sys.argv[1]
cfgs = sys.argv[2:] or ["default:"]
t0 = time.perf_counter()
base_spec, _, coarse = spec.partition("@")
p0 = make(base_spec)
problems = [(spec, p0)]
if coarse:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    lvs = oracle_py.System(p0).gamg_levels(smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
                                           mergeLevels=1)
    problems = []
    for K in coarse.split(","):
        lv = lvs[int(K) - 1]
        q = dict(nCells=lv["nCells"], lowerAddr=lv["lowerAddr"], upperAddr=lv["upperAddr"], diag=lv["diag"], upper=lv["upper"])
        q["source"] = cases.amul(q, np.sin(1e-3 * np.arange(q["nCells"])))
        q["psi"] = np.zeros(q["nCells"])
        problems.append((base_spec + "@" + K, q))
dev = torch.device("cuda", 0)
L = capi.lib()
set_before = set()
for spec, p in problems:
    nC, nF = p["nCells"], p["lowerAddr"].size
    print("mesh %s: %d cells %d faces (%.1f s)" % (spec, nC, nF, time.perf_counter() - t0), flush=True)
    d_src = torch.from_numpy(p["source"]).to(dev)
    for cfg in cfgs:
        name, _, envs = cfg.partition(":")
        for k in set_before:
            os.environ.pop(k, None)
        set_before = set()
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            os.environ[k] = v
            set_before.add(k)
        ctx = capi.Context(0)
        ts = time.perf_counter()
        a, m = capi.from_problem(ctx, p)
        info = a.info()
        NS = int(os.environ.get("PROBE_SWEEPS", "2"))
        d_psi = torch.zeros(nC, dtype=torch.float64, device=dev)
        d_w = torch.zeros(nC, dtype=torch.float64, device=dev)
        res = {}
        for label, fn in (("gs1", lambda: L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), 1)),
                          ("gs%d" % NS, lambda: L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), NS)),
                          ("dic", lambda: L.ldu_precondition(m.h, 2, capi._ptr(d_w), capi._ptr(d_src), 0)),
                          ("amul", lambda: L.ldu_amul(m.h, capi._ptr(d_w), capi._ptr(d_src)))):
            capi._chk(fn()); ctx.sync()
            if label == "gs1":
                setup = time.perf_counter() - ts
            reps = 3 if nC > 500000 else 40
            t1 = time.perf_counter()
            for _ in range(reps):
                capi._chk(fn())
            ctx.sync()
            res[label] = (time.perf_counter() - t1) / reps * 1e3
        eng = (a.sweep_engine(0), a.sweep_engine(1), a.sweep_engine(2))
        print("%-14s levels=%d padded=%s engines tri/gs/gsk=%s | GS1 %.3f ms (%.2f us/lvl)  GS%d %.3f ms  DIC(2 half sweeps) %.3f ms  "
              "Amul %.3f ms  fallbacks %d  setup %.1f s"
              % (name, info["nLevels"], "%.2fM entries" % (info["nEntriesPadded"] / 1e6), "/".join(e.split()[0] for e in eng), res["gs1"],
                 res["gs1"] * 1e3 / info["nLevels"], NS, res["gs%d" % NS], res["dic"], res["amul"], ctx.fallback_count(), setup),
              flush=True)
        if os.environ.get("PROBE_TRACE") and eng[2] == "blocks":
            bi = np.zeros(8, dtype=np.int64)
            L.ldu_debug_blocks_info.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
            capi._chk(L.ldu_debug_blocks_info(m.h, NS, bi.ctypes.data))
            nT = int(bi[5])
            buf = torch.zeros(nT * 8, dtype=torch.int64, device="cuda")
            L.ldu_debug_blocks_trace.argtypes = [C.c_void_p, C.c_void_p]
            capi._chk(L.ldu_debug_blocks_trace(m.h, C.c_void_p(buf.data_ptr())))
            capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), NS)); ctx.sync()
            capi._chk(L.ldu_debug_blocks_trace(m.h, None))
            T = buf.cpu().numpy().reshape(nT, 8).astype(np.float64)
            T = T[T[:, 3] > 0]
            T[:, :4] = (T[:, :4] - T[:, 0].min()) * 0.01
            print("   block engine: %d blocks x %d wavefronts, %d B of LDS, %d ghosts, %d tasks: first start %.1f, last stored %.1f us"
                  % (bi[0], bi[1], bi[2], bi[3], nT, T[:, 0].min(), T[:, 3].max()))
            for nm, d in (("loads issued", T[:, 1] - T[:, 0]), ("waiting", T[:, 2] - T[:, 1]), ("compute+store", T[:, 3] - T[:, 2])):
                print("      %-14s median %.2f  mean %.2f  p90 %.2f  max %.2f us  (sum %.0f us)" % (nm, np.median(d), d.mean(), np.percentile(d, 90), d.max(), d.sum()))
            for tl in (1, 2, 4, 8):
                sel = T[:, 7] == tl
                if sel.any():
                    print("      %d lane(s) per row: %d tasks, compute+store median %.2f mean %.2f us" % (tl, sel.sum(), np.median((T[:, 3] - T[:, 2])[sel]), (T[:, 3] - T[:, 2])[sel].mean()))
            for j in range(NS):
                sel = T[:, 5] == j
                print("      sweep %d: first task starts %.1f us, first stored %.1f, last stored %.1f us (%d tasks)"
                      % (j, T[sel, 0].min(), T[sel, 3].min(), T[sel, 3].max(), sel.sum()))
            # busiest wavefront: what its time is made of
            key = T[:, 6] * 16 + T[:, 4]
            ks, inv = np.unique(key, return_inverse=True)
            busy = np.bincount(inv, weights=T[:, 3] - T[:, 0])
            cnt = np.bincount(inv)
            print("      per wavefront: tasks median %d max %d; time in tasks median %.0f max %.0f us" % (np.median(cnt), cnt.max(), np.median(busy), busy.max()))
        elif os.environ.get("PROBE_TRACE") and eng[2].startswith("one"):
            nT = NS * info["nSlices"]
            buf = torch.zeros(nT * 8, dtype=torch.int64, device="cuda")
            L.ldu_debug_gs_multi_trace.argtypes = [C.c_void_p, C.c_void_p]
            capi._chk(L.ldu_debug_gs_multi_trace(m.h, C.c_void_p(buf.data_ptr())))
            capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), NS)); ctx.sync()
            capi._chk(L.ldu_debug_gs_multi_trace(m.h, None))
            T = buf.cpu().numpy().reshape(nT, 8).astype(np.float64)
            T[:, :4] = (T[:, :4] - T[:, 0].min()) * 0.01
            print("   one-workgroup engine, %d tasks: first start %.1f, last stored %.1f us" % (nT, T[:, 0].min(), T[:, 3].max()))
            for nm, d in (("loads issued", T[:, 1] - T[:, 0]), ("waiting", T[:, 2] - T[:, 1]), ("compute+store", T[:, 3] - T[:, 2])):
                print("      %-14s median %.2f  mean %.2f  p90 %.2f  max %.2f us" % (nm, np.median(d), d.mean(), np.percentile(d, 90), d.max()))
            for j in range(NS):
                sel = T[:, 5] == j
                print("      sweep %d: first task starts %.1f us, first stored %.1f, last stored %.1f us (%d tasks)"
                      % (j, T[sel, 0].min(), T[sel, 3].min(), T[sel, 3].max(), sel.sum()))
        elif os.environ.get("PROBE_TRACE") and "clusters" not in eng[2]:
            lev = np.zeros(info["nLevels"] + 8, dtype=np.int32)
            L.ldu_debug_slice_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
            capi._chk(L.ldu_debug_slice_levels(m.h, lev.ctypes.data, lev.size))
            nLev = int(lev[0]); start = lev[1:2 + nLev]; nS = int(start[-1])
            buf = torch.zeros(NS * nS * 8, dtype=torch.int64, device="cuda")
            L.ldu_debug_gs_multi_trace.argtypes = [C.c_void_p, C.c_void_p]
            capi._chk(L.ldu_debug_gs_multi_trace(m.h, C.c_void_p(buf.data_ptr())))
            capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), NS)); ctx.sync()
            capi._chk(L.ldu_debug_gs_multi_trace(m.h, None))
            T = buf.cpu().numpy().reshape(NS, nS, 8).astype(np.float64)
            if os.environ.get("PROBE_DUMP"):
                sl = np.zeros((nS, 5), dtype=np.int32)
                L.ldu_debug_slices.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
                capi._chk(L.ldu_debug_slices(m.h, sl.ctypes.data, nS))
                np.savez_compressed(os.environ["PROBE_DUMP"] + "_" + name + ".npz", T=buf.cpu().numpy().reshape(NS, nS, 8)[:, :, :6],
                                    start=start, slices=sl)
            if not (T[:, :, 0] > 0).any():
                m.close(); a.close(); ctx.close()
                continue
            tmin = T[:, :, 0][T[:, :, 0] > 0].min()
            T[:, :, :4] = (T[:, :, :4] - tmin) * 0.01
            for j in range(NS):
                X = T[j]
                done = np.array([X[start[l]:start[l + 1], 3].max() for l in range(nLev)])
                d = np.diff(done)
                print("   sweep %d: level 0 done at %.0f us, last at %.0f us; per level mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.1f us"
                      % (j, done[0], done[-1], d.mean(), *np.percentile(d, [50, 90, 99]), d.max()))
                print("      per task medians: start->upper %.2f  upper->ready %.2f  ready->stored %.2f us; p99: %.2f %.2f %.2f"
                      % (np.median(X[:, 1] - X[:, 0]), np.median(X[:, 2] - X[:, 1]), np.median(X[:, 3] - X[:, 2]),
                         np.percentile(X[:, 1] - X[:, 0], 99), np.percentile(X[:, 2] - X[:, 1], 99),
                         np.percentile(X[:, 3] - X[:, 2], 99)))
                # which slice finishes a level last, and how wide is it
                q = max(1, nLev // 10)
                print("      level: slices, done [us]:", "  ".join("%d: %d, %.0f" % (l, start[l + 1] - start[l], done[l]) for l in range(0, nLev, q)))
        m.close(); a.close(); ctx.close()

