import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
import numpy as np, ctypes as C
p = cases.random_graph(200000, avg_deg=9, band=3000, seed=3)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
x = m.smooth("GaussSeidel", p["psi"], p["source"], 2)
try:
    m.gSumMag(x)
    print("no abort")
except Exception as e:
    print(str(e)[:50])
n = p["nCells"]
tags = np.zeros(n, dtype=np.int32); lev = np.zeros(n, dtype=np.int32)
capi.lib().ldu_debug_granule_tags(m.h, capi._ptr(tags), capi._ptr(lev))
print("tag counts", {int(t): int((tags == t).sum()) for t in np.unique(tags)})
for t in np.unique(tags):
    L = lev[tags == t]
    print(" tag", t, "levels min/max", L.min(), L.max())
# first level that is not completely at tag>=1 (sweep 0 done) etc
nl = lev.max() + 1
for t in (1, 2):
    full = [(lev[tags >= t] == L).sum() == (lev == L).sum() for L in range(nl)]
    firstbad = next((L for L in range(nl) if not full[L]), None)
    print(" first level not fully >= tag", t, ":", firstbad, "rows in that level", (lev == firstbad).sum() if firstbad is not None else 0,
          "done", ((lev == firstbad) & (tags >= t)).sum() if firstbad is not None else 0)
    if firstbad is not None:
        idx = np.nonzero((lev == firstbad) & (tags < t))[0][:5]
        print("   sample stuck rows (level order idx)", idx)
l, u = p["lowerAddr"], p["upperAddr"]
lvl = np.zeros(n, dtype=np.int64)
for f in range(l.size):
    lvl[u[f]] = max(lvl[u[f]], lvl[l[f]] + 1)
order = np.argsort(lvl, kind="stable")
iperm = np.empty(n, dtype=np.int64); iperm[order] = np.arange(n)
assert np.array_equal(lvl[order], lev)
for r in idx[:3] if 'idx' in dir() else []:
    pass
fb = next(L for L in range(nl) if ((lev == L) & (tags < 1)).any())
stuck0 = np.nonzero((lev == fb) & (tags < 1))[0][:6]
print('first bad level', fb)
for r in stuck0:
    c = order[r]
    lows = l[u == c]; ups = u[l == c]
    print("row", r, "slice-lane", r - np.nonzero(lev == fb)[0][0], "cell", c, "nL", lows.size, "nU", ups.size, "lower nbrs new idx", iperm[lows], "levels", lvl[lows], "tags", tags[iperm[lows]])
    print("      upper nbrs levels", lvl[ups], "tags", tags[iperm[ups]])
# violations: a row at tag 2 (sweep 1 done) with an upper neighbour not yet at tag >= 1
t_new = tags[iperm]          # tags per old cell
viol = np.nonzero((t_new[l] >= 2) & (t_new[u] < 1))[0]
print("faces with lower side at tag 2 but upper side never done:", viol.size)
if viol.size:
    f = viol[0]
    print("  e.g. lower cell", l[f], "level", lvl[l[f]], "nU", (l == l[f]).sum(), "nL", (u == l[f]).sum(), "upper cell", u[f], "level", lvl[u[f]],
          "position of this face among the owner's faces", int(np.nonzero(np.nonzero(l == l[f])[0] == f)[0][0]))
stuck_all = np.nonzero((lev == 70) & (tags < 1))[0]
cnt2 = 0
for r in stuck_all:
    c = order[r]
    lows = l[u == c]
    if (tags[iperm[lows]] == 2).any(): cnt2 += 1
print("stuck level-70 rows with a lower nbr already at tag 2:", cnt2, "of", stuck_all.size)
rec = np.zeros(1 + 512, dtype=np.int32)
capi.lib().ldu_debug_p2p_records(m.h, capi._ptr(rec))
nrec = min(int(rec[0]), 64)
print("expired lower-dependency waits recorded:", rec[0])
R = rec[1:].reshape(64, 8)[:nrec]
for r_ in R[:12]:
    kind, row, tag, col, sy, sw, k_, n_ = r_
    print("  row %d (level %d) waits tag %d on col %d (level %d): wave saw tags (%d,%d); memory now has tag %d"
          % (row, lev[row], tag, col, lev[col], sy, sw, tags[col]))
