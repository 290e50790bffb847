"""Probe: time GaussSeidel / DIC sweeps and Amul on the n^3 box for several engine settings.
usage: python tools/sweep_probe.py [n] [cfg ...]   cfg = name:ENV=VAL,ENV=VAL"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
from openfoam_amd import capi, cases
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
cfgs = sys.argv[2:] or ["p2p4:LDU_SWEEP=p2p,LDU_P2P_BPC=4", "levels:LDU_SWEEP=levels"]
p = cases.box3d(n)
nC, nF = p["nCells"], p["lowerAddr"].size
dev = torch.device("cuda", 0)
d_src = torch.from_numpy(p["source"]).to(dev)
for cfg in cfgs:
    name, _, envs = cfg.partition(":")
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    ctx = capi.Context(0)
    a, m = capi.from_problem(ctx, p)
    info = a.info()
    d_psi = torch.zeros(nC, dtype=torch.float64, device=dev)
    d_w = torch.zeros(nC, dtype=torch.float64, device=dev)
    import ctypes as C
    L = capi.lib()
    # warm
    NS = int(os.environ.get("PROBE_SWEEPS", "4"))
    capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), NS))
    capi._chk(L.ldu_precondition(m.h, 2, capi._ptr(d_w), capi._ptr(d_src), 0))
    m.profile_begin()
    t0 = time.perf_counter()
    capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), NS))
    t1 = time.perf_counter()
    for _ in range(3):
        capi._chk(L.ldu_precondition(m.h, 2, capi._ptr(d_w), capi._ptr(d_src), 0))
    prof = m.profile_end()
    if "gs_multi" in prof:
        gs = prof["gs_multi"]["ms"] / prof["gs_multi"]["count"] / float(NS)
    else:
        gs = prof["gs_sweep"]["ms"] / prof["gs_sweep"]["count"]
    tri = prof["tri_sweep"]["ms"] / prof["tri_sweep"]["count"]
    print("%-12s n=%d levels=%d  GS sweep %.3f ms (%.2f us/level, %.0f GB/s alg)  DIC half-sweep %.3f ms  (0 GS wall %.1f ms)"
          % (name, n, info["nLevels"], gs, gs * 1e3 / info["nLevels"], (60.0 * nC + 12.0 * nF) / gs / 1e6, tri,
             (t1 - t0) * 1e3), "sweeps/launch", NS, flush=True)
    m.close(); a.close(); ctx.close()
