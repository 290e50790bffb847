"""Trace one GaussSeidel sweep of the point-to-point engine on the n^3 box and print where the
time goes: per-slice wait / poll statistics and the per-level hand-off latency."""
import os
import sys
import ctypes as C

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
from openfoam_amd import capi, cases
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
os.environ["LDU_GS_PIPELINE"] = "0"
p = cases.box3d(n)
nC = p["nCells"]
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
info = a.info()
nS = info["nSlices"]
d_src = torch.from_numpy(p["source"]).to(dev)
d_psi = torch.zeros(nC, dtype=torch.float64, device=dev)
L = capi.lib()
capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), 1))
trace = torch.zeros(nS * 8, dtype=torch.int64, device=dev)
capi._chk(L.ldu_debug_p2p_trace(m.h, capi._ptr(trace)))
capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_psi), capi._ptr(d_src), 1))
capi._chk(L.ldu_debug_p2p_trace(m.h, None))
T = trace.cpu().numpy().reshape(nS, 8)
tT, tW, tR, tD, polls, xcc, blk, chunk = [T[:, i] for i in range(8)]
tT, tW, tR, tD = [x * 10 for x in (tT, tW, tR, tD)]   # wall_clock64: 100 MHz -> ns
t0 = tT.min()
clk = 2.4e3  # ns per us (s_memtime = shader clock; only ratios matter)
tot = (tD.max() - t0)
print("slices %d  levels %d  sweep span %.1f us" % (nS, info["nLevels"], tot / 1e3))
print("per slice [ns]: ticket->waitStart median %.0f  wait median %.0f (p10 %.0f p90 %.0f)  ready->done median %.0f"
      % (np.median(tW - tT), np.median(tR - tW), np.percentile(tR - tW, 10), np.percentile(tR - tW, 90),
         np.median(tD - tR)))
print("polls per slice: median %.0f mean %.1f max %d" % (np.median(polls), polls.mean(), polls.max()))
# level structure: slices are in level order; recover level boundaries from the plan via rows
# approximate levels by the cube formula
lev_sizes = []
N = n
for Lv in range(3 * N - 2):
    lo = max(0, Lv - 2 * (N - 1)); cnt = 0
    # number of (i,j,k) with i+j+k = Lv, 0<=i,j,k<N
    for k in range(max(0, Lv - 2 * (N - 1)), min(N - 1, Lv) + 1):
        r = Lv - k
        cnt += max(0, min(N - 1, r) - max(0, r - (N - 1)) + 1)
    lev_sizes.append(cnt)
starts = np.cumsum([0] + [(c + 63) // 64 for c in lev_sizes])
assert starts[-1] == nS, (starts[-1], nS)
done_lvl = np.array([tD[starts[i]:starts[i + 1]].max() for i in range(len(lev_sizes))])
ready_lvl = np.array([tR[starts[i]:starts[i + 1]].max() for i in range(len(lev_sizes))])
first_ready = np.array([tR[starts[i]:starts[i + 1]].min() for i in range(len(lev_sizes))])
hop = np.diff(done_lvl)
print("level completion spacing [ns]: median %.0f  mean %.0f  (first 50 levels mean %.0f, middle mean %.0f)"
      % (np.median(hop), hop.mean(), hop[:50].mean(), hop[250:400].mean()))
# hand-off latency: consumer slice ready time minus the latest done time in the previous level
lat = first_ready[1:] - np.array([tD[starts[i]:starts[i + 1]].min() for i in range(len(lev_sizes) - 1)])
print("earliest ready(L+1) - earliest done(L) [ns]: median %.0f" % np.median(lat))
mid = slice(starts[300], starts[301])
print("level 300: %d slices; done spread %.0f ns; XCCs used %s; polls mean %.1f"
      % (starts[301] - starts[300], tD[mid].max() - tD[mid].min(), np.unique(xcc[mid]), polls[mid].mean()))
# how far ahead do waves take tickets? ticket time vs ready time
ahead = (tR - tT)
print("ticket->ready [ns]: median %.0f p90 %.0f  => window of %.1f levels at the median level spacing"
      % (np.median(ahead), np.percentile(ahead, 90), np.median(ahead) / max(np.median(hop), 1)))
