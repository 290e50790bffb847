"""which wait expires in the chip-wide pipelined GaussSeidel kernel at large upper-neighbour skew?  Prints the records the
aborting lanes leave (row, column waited for, expected tag, tags seen) with the dependency levels of both rows."""
import ctypes as C
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch
torch.cuda.init()
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = cases.irregular_box(n)
order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
p = cases.renumbered(p, order, fmap, flip, nl, nu)
ctx = capi.Context(0)
ctx.set_spin_limit(int(os.environ.get("PROBE_SPIN", "200000")))
a, m = capi.from_problem(ctx, p)
print("info", a.info(), [a.sweep_engine(q) for q in (0, 1, 2)], flush=True)
rng = np.random.RandomState(1)
x, b = rng.randn(p["nCells"]), rng.randn(p["nCells"])
f0 = ctx.fallback_count()
m.smooth("GaussSeidel", x, b, k)
print("fallbacks", ctx.fallback_count() - f0)
rec = (C.c_int32 * (1 + 64 * 8))()
capi.lib().ldu_debug_p2p_records(m.h, rec)
nrec = rec[0]
tags = np.zeros(p["nCells"], dtype=np.int32); lev = np.zeros(p["nCells"], dtype=np.int32)
capi.lib().ldu_debug_granule_tags(m.h, capi._ptr(tags), capi._ptr(lev))
print("records", nrec)
for t in range(0, 4):
    sel = tags == t
    if sel.any():
        print("tag %d: %d rows, levels %d .. %d" % (t, sel.sum(), lev[sel].min(), lev[sel].max()))
for L in range(260, 275):
    sel = lev == L
    print("level", L, "rows", sel.sum(), "tags", {int(t): int((tags[sel] == t).sum()) for t in np.unique(tags[sel])})
for i in list(range(0, 10)) + list(range(32, 40)):
    r = rec[1 + 8 * i: 1 + 8 * i + 8]
    kind, row, exp, col, sy, sw, aux = r[:7]
    if r[7] == 0: continue
    print("kind %d (1 lower / 2 upper) row %d (level %d) waits for col %d (level %d) expected tag %d seen %d/%d entry %d" %
          (kind, row, lev[row], col & 0x7fffffff, lev[col & 0x7fffffff], exp, sy, sw, aux))

# host-side reconstruction: rows are ordered by (dependency level, original index)
l, u = p["lowerAddr"], p["upperAddr"]
nC = p["nCells"]
level = np.zeros(nC, dtype=np.int64)
for f in range(l.size):
    if level[u[f]] < level[l[f]] + 1: level[u[f]] = level[l[f]] + 1
order_rows = np.lexsort((np.arange(nC), level))        # row -> cell
row_of = np.empty(nC, dtype=np.int64); row_of[order_rows] = np.arange(nC)
assert np.array_equal(level[order_rows], lev), "level order reconstruction"
tag_cell = tags[row_of]
front = lev[tags == 0].min()
stuck = np.nonzero((level == front) & (tag_cell == 0))[0]
print("sweep-0 frontier level", front, "stuck cells", stuck.size)
lo_start = np.searchsorted(u[np.argsort(u, kind="stable")], np.arange(nC + 1))
lo_faces = np.argsort(u, kind="stable")
for c in stuck[:6]:
    fs = lo_faces[lo_start[c]:lo_start[c + 1]]
    print(" cell", c, "row", row_of[c], "lower nbrs (cell, level, tag):", [(int(l[f]), int(level[l[f]]), int(tag_cell[l[f]])) for f in fs])
    ufs = np.nonzero(l == c)[0]
    print("    upper nbrs (cell, level, tag):", [(int(u[f]), int(level[u[f]]), int(tag_cell[u[f]])) for f in ufs])
