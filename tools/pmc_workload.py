"""Small fixed workload for PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), driven by tools/pmc_traffic.py:
3x {dot product, reciprocal (known bytes: the two calibration kernels), Amul, k pipelined GaussSeidel sweeps, DIC apply}
on one mesh.  usage: python tools/pmc_workload.py [MESH=box:216] [k=2]   (MESH as in tools/mesh_probe.py)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
from openfoam_amd import capi, cases, octree
import torch


def make(spec):
    f = spec.split(":")
    if f[0] == "box":
        return cases.box3d(int(f[1]))
    if f[0] == "irregular":
        p = cases.irregular_box(int(f[1]))
    elif f[0] == "motorbike":
        from openfoam_amd import motorbike
        p = motorbike.problem(f[1])
        p.pop("cellLevel"); p.pop("meta")
        if not (len(f) > 2 and f[2] == "rcm"):
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


spec = sys.argv[1] if len(sys.argv) > 1 else "box:216"
if spec.isdigit():
    spec = "box:" + spec
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = make(spec)
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
nC = p["nCells"]
d_src = torch.from_numpy(p["source"]).to(dev)
d_x = torch.zeros(nC, dtype=torch.float64, device=dev)
d_y = torch.zeros(nC, dtype=torch.float64, device=dev)
L = capi.lib()
r = C.c_double()
for _ in range(3):
    capi._chk(L.ldu_gSumProd(m.h, capi._ptr(d_src), capi._ptr(d_src), C.byref(r)))
    capi._chk(L.ldu_amul(m.h, capi._ptr(d_y), capi._ptr(d_src)))
    capi._chk(L.ldu_smooth(m.h, 0, capi._ptr(d_x), capi._ptr(d_src), k))
    capi._chk(L.ldu_precondition(m.h, 2, capi._ptr(d_y), capi._ptr(d_src), 0))
print("pmc_workload %s: nCells %d nFaces %d engine %s done %g" % (spec, nC, p["lowerAddr"].size, a.sweep_engine(2), r.value))
m.close(); a.close(); ctx.close()
