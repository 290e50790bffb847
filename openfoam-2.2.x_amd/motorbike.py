"""The p-equation matrix on a REAL snappyHexMesh motorBike mesh (the workload BASELINE's metric is named after;
SURVEY.md 8f rank 3).

The mesh is made by the reference's own blockMesh + snappyHexMesh (castellatedMesh only) from the reference's own
motorBike.obj (oracle/motorbike_case.py, oracle/build_ref_mesh.sh, tools/make_motorbike.py - where /root/reference
exists) and stored in a compact form under data/motorbike/<name>.npz (not in git): lduAddressing of the internal faces
(owner counts + neighbour labels, the cell numbering hexRef8 / snappyHexMesh produced), the face normal direction, the
refinement level of every cell and the cells of the outlet patch.  tools/make_motorbike.py verified against the mesh's own
geometry (the reference's face / cell formulas) that every cell is a cube of its level and every internal face an
axis-aligned square of the finer cell, so |Sf| and the normal distance n.d between the cell centres - all the Laplacian
needs (gaussLaplacianScheme.C:43-114 with nonOrthDeltaCoeffs, surfaceInterpolation.C:252-300) - follow from the levels to
1e-13.

problem(): simpleFoam's pEqn on it, as for the octree twin and the box stand-in: upper = -rAUf |Sf| /
(n.d) with rAUf = 1 + 0.5 u01(seed, f), diag = negSumDiag + the fixedValue outlet's internalCoeffs (zeroGradient on walls,
inlet and the bike, motorBike/0.org/p), b = A x* for a spatially smooth x* (a random field averaged over the face
neighbours), faceAreaPair weights by the reference's formula (faceAreaPairGAMGAgglomeration.C:59-72)."""
import json
import os

import numpy as np

from . import cases as _cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STORE = os.path.join(ROOT, "data", "motorbike")


def path(name):
    return os.path.join(STORE, name + ".npz")


def available(name):
    return os.path.exists(path(name))


def load(name):
    """-> dict(nCells, lowerAddr, upperAddr, dirs, level, outletCells, h0, meta)"""
    if not available(name):
        raise FileNotFoundError("%s: no such motorBike mesh (made by tools/make_motorbike.py where /root/reference exists)" % path(name))
    z = np.load(path(name))
    meta = json.loads(bytes(z["meta"]).decode())
    cnt = z["ownerCount"].astype(np.int64)
    nC = cnt.size
    lower = np.repeat(np.arange(nC, dtype=np.int32), cnt)
    upper = z["upper"].astype(np.int32)
    if lower.size != upper.size or lower.size != meta["nInternalFaces"]:
        raise ValueError("%s: inconsistent face counts" % path(name))
    return dict(nCells=nC, lowerAddr=lower, upperAddr=upper, dirs=z["dirs"], level=z["cellLevel"],
                outletCells=z["outletCells"], h0=float(meta["h0"]), meta=meta)


def decomposition(name, n_ranks):
    """rank of every cell for an n_ranks-way run: the reference's own `hierarchical` decomposition of the cell centres
    (decomposeParDict of the tutorial; made by tools/make_motorbike.py through oracle/decomp_driver.C) - None when not stored"""
    f = os.path.join(STORE, name + "_decomp.npz")
    if not os.path.exists(f):
        return None
    z = np.load(f)
    key = "proc%d" % n_ranks
    return z[key].astype(np.int64) if key in z else None


def smooth_field(nC, l, u, seed=777, passes=12):
    """a spatially smooth field whatever the cell numbering: u01 noise relaxed `passes` times towards the average of the face
    neighbours (x <- (x + mean of neighbours) / 2)"""
    x = _cases.u01(seed, nC) - 0.5
    deg = (np.bincount(l, minlength=nC) + np.bincount(u, minlength=nC)).astype(np.float64)
    deg[deg == 0] = 1.0
    for _ in range(passes):
        s = np.bincount(l, weights=x[u], minlength=nC) + np.bincount(u, weights=x[l], minlength=nC)
        x = 0.5 * (x + s / deg)
    return x / np.abs(x).max()


def problem(name="mb12", seed=12345, m=None):
    if m is None:
        m = load(name)
    l, u, lvl = m["lowerAddr"], m["upperAddr"], m["level"].astype(np.int64)
    nC, nF = m["nCells"], l.size
    lf = np.maximum(lvl[l], lvl[u])
    hf = m["h0"] / (1 << lf)
    area = hf * hf
    dn = np.where(lvl[l] != lvl[u], 1.5 * hf, hf)          # h_fine / 2 + h_coarse / 2
    hmin = m["h0"] / (1 << int(lvl.max()))
    upper = -(1.0 + 0.5 * _cases.u01(seed, nF)) * (area / dn) / hmin
    diag = _cases._neg_sum_diag(nC, l, u, upper, upper)
    oc = m["outletCells"].astype(np.int64)
    h = m["h0"] / (1 << lvl[oc])
    np.add.at(diag, oc, (h * h / (0.5 * h)) / hmin)        # fixedValue: gamma |Sf| deltaCoeffs at the patch face (gamma = 1)
    comp = (area / np.sqrt(area)) * np.array([1.0, 1.01, 1.02])[m["dirs"]]
    p = dict(nCells=nC, lowerAddr=l, upperAddr=u, upper=upper, diag=diag, faceWeights=np.sqrt(comp * comp))
    p["source"] = _cases.amul(p, smooth_field(nC, l, u))
    p["psi"] = np.zeros(nC)
    p["cellLevel"] = m["level"]
    p["meta"] = m["meta"]
    return p


def dumped_problem(name="mb2sl_p3"):
    """the p-matrix of a real SIMPLE iteration of the REFERENCE's simpleFoam on a snapped + layered motorBike mesh (polyhedral
    cells, layer prisms, non-orthogonal faces), as tools/make_motorbike_matrix.py stored it: `laplacian((1|A(U)),p)` with its
    boundary coefficients (pEqn.H:14-21; negative definite: upper > 0, diag < 0, gaussLaplacianScheme.C:57-73), its source, and
    the faceAreaPair weights of the mesh's face area vectors.  psi = 0 (the bench's convention)."""
    if not available(name):
        raise FileNotFoundError("%s: no such stored matrix (made by tools/make_motorbike_matrix.py where /root/reference exists)" % path(name))
    z = np.load(path(name))
    meta = json.loads(bytes(z["meta"]).decode())
    cnt = z["ownerCount"].astype(np.int64)
    nC = cnt.size
    return dict(nCells=nC, lowerAddr=np.repeat(np.arange(nC, dtype=np.int32), cnt), upperAddr=z["upperAddr"].astype(np.int32),
                diag=z["diag"].astype(np.float64), upper=z["upper"].astype(np.float64), source=z["source"].astype(np.float64),
                psi=np.zeros(nC), faceWeights=z["faceWeights"].astype(np.float64), meta=meta)
