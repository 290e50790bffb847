import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases, motorbike
import oracle_py
name = sys.argv[1]; rcm = int(sys.argv[2])
t0 = time.time()
p = motorbike.problem(name)
p.pop("cellLevel"); p.pop("meta")
if rcm:
    order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
    nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
    p = cases.renumbered(p, order, fmap, flip, nl, nu)
print("problem", time.time() - t0, p["nCells"], flush=True)
S = oracle_py.System(p)
lv = S.gamg_levels(smoother="GaussSeidel", nCellsInCoarsestLevel=10, mergeLevels=1, agglomerator="faceAreaPair")
print("levels", time.time() - t0, len(lv), flush=True)
out = "/tmp/bp/%s_%d" % (name, rcm)
os.makedirs(out, exist_ok=True)
def dump(i, nC, l, u):
    with open("%s/L%d.bin" % (out, i), "wb") as f:
        np.array([nC, l.size], dtype=np.int32).tofile(f); l.astype(np.int32).tofile(f); u.astype(np.int32).tofile(f)
dump(0, p["nCells"], p["lowerAddr"], p["upperAddr"])
for i, L in enumerate(lv):
    dump(i + 1, L["nCells"], L["lowerAddr"], L["upperAddr"])
    print(i + 1, L["nCells"], L["lowerAddr"].size)
