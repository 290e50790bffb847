"""Octree-castellated hex mesh -> lduMatrix problem: the topological twin of a snappyHexMesh "castellated" mesh.

What the reference's mesher produces for the motorBike case:
  blockMesh background box 20x8x8 over (-5 -4 0)..(15 4 8)   tutorials/incompressible/simpleFoam/motorBike/constant/polyMesh/blockMeshDict
  snappyHexMesh castellation                                  system/snappyHexMeshDict:64-162
    - cells cut by the surface refined to level 5, to level 6 where the surface is sharply curved / on feature edges
      (`level (5 6)`, `features level 6`, `resolveFeatureAngle 30`)
    - `refinementBox` (-1 -0.7 0)..(8 0.7 2.5) refined to level 4
    - `nCellsBetweenLevels 3` buffer layers, 2:1 balance across faces (hexRef8::consistentRefinement,
      src/dynamicMesh/polyTopoChange/polyTopoChange/hexRef8.C)
    - every refinement pass splits the marked hexes in 8: the parent keeps its cell label, the 7 other children are
      appended behind all existing cells in ascending parent order (hexRef8::setRefinement: `cAdded[0] = cellI`,
      `polyAddCell` for 1..7)
    - cells inside the body are removed, the labels compacted in order (meshRefinement::splitMeshRegions / removeCells)
    - faces come out in upper-triangular order (polyTopoChange::compactAndReorder)
  A hex next to four finer hexes sees four quarter faces there ("hanging faces"): rows with up to 24 neighbours.

Not restated: the geometry (the tutorial's motorBike.obj.gz does not exist on the GPU box, so the body here is an analytic
union of wheels / frame / rider primitives of the same overall size), the exact child order inside a split hex (hexRef8 orders
the children by the cell's anchor points; here: x fastest) and the snapping / layer phases (they move points, not topology).
The result is the matrix graph the solver sees: octree hexes with hanging faces in hexRef8's numbering.

Everything is numpy on integer cell coordinates; no reference file is read.
"""
import numpy as np

from . import cases as _cases

_DIRS = ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))


# ---------------------------------------------------------------------------------------------------------------
# the body: signed distances (negative inside) of simple primitives; metres, motorBike-sized
# ---------------------------------------------------------------------------------------------------------------
def _sd_sphere(x, y, z, c, r):
    return np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - r


def _sd_capsule(x, y, z, a, b, r):
    ax, ay, az = a
    bx, by, bz = b
    dx, dy, dz = bx - ax, by - ay, bz - az
    L2 = dx * dx + dy * dy + dz * dz
    t = np.clip(((x - ax) * dx + (y - ay) * dy + (z - az) * dz) / L2, 0.0, 1.0)
    return np.sqrt((x - ax - t * dx) ** 2 + (y - ay - t * dy) ** 2 + (z - az - t * dz) ** 2) - r


def _sd_wheel(x, y, z, c, R, w):
    """disc of radius R and half width w, axis along y"""
    rr = np.sqrt((x - c[0]) ** 2 + (z - c[2]) ** 2) - R
    yy = np.abs(y - c[1]) - w
    outside = np.sqrt(np.maximum(rr, 0.0) ** 2 + np.maximum(yy, 0.0) ** 2)
    return outside + np.minimum(np.maximum(rr, yy), 0.0)


def _sd_ellipsoid(x, y, z, c, s):
    """approximate signed distance of an ellipsoid (exact on the axes; good enough to mark cells)"""
    k0 = np.sqrt(((x - c[0]) / s[0]) ** 2 + ((y - c[1]) / s[1]) ** 2 + ((z - c[2]) / s[2]) ** 2)
    k1 = np.sqrt(((x - c[0]) / s[0] ** 2) ** 2 + ((y - c[1]) / s[1] ** 2) ** 2 + ((z - c[2]) / s[2] ** 2) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.where(k1 > 0, k0 * (k0 - 1.0) / k1, -min(s))
    return d


def body_distance(x, y, z):
    """(distance to the whole body, distance to its sharply curved parts = the level-6 regions)"""
    wheels = np.minimum(_sd_wheel(x, y, z, (0.02, 0.0, 0.33), 0.32, 0.07),
                        _sd_wheel(x, y, z, (1.45, 0.0, 0.33), 0.32, 0.08))
    head = _sd_sphere(x, y, z, (0.78, 0.0, 1.38), 0.13)
    fork = _sd_capsule(x, y, z, (0.05, 0.0, 0.35), (0.42, 0.0, 1.02), 0.05)
    bars = _sd_capsule(x, y, z, (0.42, -0.36, 1.04), (0.42, 0.36, 1.04), 0.035)
    frame = _sd_ellipsoid(x, y, z, (0.80, 0.0, 0.62), (0.62, 0.19, 0.30))
    torso = _sd_capsule(x, y, z, (1.05, 0.0, 0.92), (0.80, 0.0, 1.22), 0.17)
    arms = np.minimum(_sd_capsule(x, y, z, (0.82, -0.22, 1.20), (0.45, -0.33, 1.05), 0.05),
                      _sd_capsule(x, y, z, (0.82, 0.22, 1.20), (0.45, 0.33, 1.05), 0.05))
    legs = np.minimum(_sd_capsule(x, y, z, (1.05, -0.2, 0.85), (0.75, -0.24, 0.40), 0.07),
                      _sd_capsule(x, y, z, (1.05, 0.2, 0.85), (0.75, 0.24, 0.40), 0.07))
    sharp = np.minimum(np.minimum(wheels, head), np.minimum(np.minimum(fork, bars), arms))
    whole = np.minimum(np.minimum(sharp, frame), np.minimum(torso, legs))
    return whole, sharp


def _sd_box(x, y, z, lo, hi):
    qx = np.maximum(lo[0] - x, x - hi[0])
    qy = np.maximum(lo[1] - y, y - hi[1])
    qz = np.maximum(lo[2] - z, z - hi[2])
    out = np.sqrt(np.maximum(qx, 0) ** 2 + np.maximum(qy, 0) ** 2 + np.maximum(qz, 0) ** 2)
    return out + np.minimum(np.maximum(qx, np.maximum(qy, qz)), 0.0)


# ---------------------------------------------------------------------------------------------------------------
class _Index:
    """(level, i, j, k) -> cell label through one sorted key array"""

    def __init__(self, base, maxLevel):
        self.nx = [base[0] << L for L in range(maxLevel + 2)]
        self.ny = [base[1] << L for L in range(maxLevel + 2)]
        self.nz = [base[2] << L for L in range(maxLevel + 2)]
        self.X = np.int64(self.nx[maxLevel + 1] + 2)
        self.Y = np.int64(self.ny[maxLevel + 1] + 2)
        self.Z = np.int64(self.nz[maxLevel + 1] + 2)

    def key(self, lvl, i, j, k):
        return ((lvl.astype(np.int64) * self.Z + (k + 1)) * self.Y + (j + 1)) * self.X + (i + 1)

    def build(self, lvl, i, j, k):
        keys = self.key(lvl, i, j, k)
        self.order = np.argsort(keys, kind="stable")
        self.sorted = keys[self.order]

    def find(self, lvl, i, j, k):
        """labels of the cells (lvl,i,j,k); -1 where no such leaf exists"""
        q = self.key(lvl, i, j, k)
        pos = np.searchsorted(self.sorted, q)
        pos[pos >= self.sorted.size] = 0
        hit = self.sorted[pos] == q
        return np.where(hit, self.order[pos], -1)


def generate(base=(60, 24, 24), lo=(-5.0, -4.0, 0.0), size=20.0, surface_levels=(5, 6), box_level=4,
             box=((-1.0, -0.7, 0.0), (8.0, 0.7, 2.5)), n_buffer=3, verbose=False):
    """The leaf cells of the castellated octree in hexRef8's numbering.

    base: background cells (the tutorial: 20x8x8; 60x24x24 = the same box three times finer gives ~10 M cells with
    the tutorial's own levels).  Returns dict(level int8[nC], i, j, k int32[nC] = integer coordinates at the cell's
    own level, h0 = background cell size)."""
    h0 = size / base[0]
    maxLevel = max(surface_levels[1], box_level)
    idx = _Index(base, maxLevel)
    c = np.arange(base[0] * base[1] * base[2], dtype=np.int64)
    I = (c % base[0]).astype(np.int32)
    J = ((c // base[0]) % base[1]).astype(np.int32)
    K = (c // (base[0] * base[1])).astype(np.int32)
    lvl = np.zeros(c.size, dtype=np.int8)
    hL = [h0 / (1 << L) for L in range(maxLevel + 2)]

    def centres(sel=slice(None)):
        h = h0 / (1 << lvl[sel].astype(np.int64))
        return lo[0] + (I[sel] + 0.5) * h, lo[1] + (J[sel] + 0.5) * h, lo[2] + (K[sel] + 0.5) * h, h

    # the geometric criterion is static: a cell that was not marked when it was created never will be, so every pass
    # only evaluates the cells the previous pass created (the 2:1 balance below can still drag any cell along)
    fresh = np.arange(lvl.size, dtype=np.int64)
    for it in range(4 * (maxLevel + 1)):
        x, y, z, h = centres(fresh)
        r = h * (np.sqrt(3.0) / 2.0)
        dAll, dSharp = body_distance(x, y, z)
        dAll, dSharp = np.abs(dAll), np.abs(dSharp)
        dBox = _sd_box(x, y, z, box[0], box[1])
        fl = lvl[fresh]
        want = np.zeros(fresh.size, dtype=bool)
        for L in range(maxLevel):
            atL = fl == L
            if not atL.any():
                continue
            for dist, T in ((dAll, surface_levels[0]), (dSharp, surface_levels[1]), (dBox, box_level)):
                if T > L:
                    band = (n_buffer - 1) * sum(hL[m] for m in range(L + 1, T + 1))
                    want |= atL & (dist <= r + band)
        mark = np.zeros(lvl.size, dtype=bool)
        mark[fresh[want]] = True
        # 2:1 balance across faces: a marked cell whose neighbour is one level coarser drags that neighbour along
        idx.build(lvl, I, J, K)
        front = np.flatnonzero(mark)
        while front.size:
            new = []
            fl, fi, fj, fk = lvl[front], I[front], J[front], K[front]
            for dx, dy, dz in _DIRS:
                ni, nj, nk = fi + dx, fj + dy, fk + dz
                same = idx.find(fl, ni, nj, nk)
                miss = (same < 0) & (fl > 0)
                if not miss.any():
                    continue
                par = idx.find(fl[miss] - 1, ni[miss] >> 1, nj[miss] >> 1, nk[miss] >> 1)
                par = par[par >= 0]
                par = par[~mark[par]]
                if par.size:
                    par = np.unique(par)
                    mark[par] = True
                    new.append(par)
            front = np.unique(np.concatenate(new)) if new else np.zeros(0, dtype=np.int64)
        parents = np.flatnonzero(mark)
        if verbose:
            print("octree pass %d: %d cells, %d to split" % (it, lvl.size, parents.size), flush=True)
        if parents.size == 0:
            break
        # split: child 0 stays in place, children 1..7 appended in ascending parent order
        pi, pj, pk, pl = I[parents].astype(np.int64), J[parents].astype(np.int64), K[parents].astype(np.int64), lvl[parents]
        I[parents], J[parents], K[parents] = 2 * pi, 2 * pj, 2 * pk
        lvl[parents] = pl + 1
        a = np.arange(1, 8)
        I = np.concatenate([I, (2 * pi[:, None] + (a & 1)[None, :]).ravel().astype(np.int32)])
        J = np.concatenate([J, (2 * pj[:, None] + ((a >> 1) & 1)[None, :]).ravel().astype(np.int32)])
        K = np.concatenate([K, (2 * pk[:, None] + ((a >> 2) & 1)[None, :]).ravel().astype(np.int32)])
        fresh = np.concatenate([parents, np.arange(lvl.size, lvl.size + 7 * parents.size, dtype=np.int64)])
        lvl = np.concatenate([lvl, np.repeat(pl + 1, 7).astype(np.int8)])
    # castellation: drop the cells whose centre lies inside the body, labels compacted in order
    x, y, z, h = centres()
    keep = body_distance(x, y, z)[0] > 0.0
    return dict(level=lvl[keep], i=I[keep], j=J[keep], k=K[keep], h0=h0, lo=lo, base=base, maxLevel=maxLevel)


def faces(m):
    """internal faces of the leaf cells: (lower, upper, area, normal distance between the centres, direction),
    in upper-triangular order.  A face between a cell and a coarser neighbour has the FINE cell's area."""
    lvl, I, J, K = m["level"], m["i"], m["j"], m["k"]
    nC = lvl.size
    idx = _Index(m["base"], m["maxLevel"])
    idx.build(lvl, I, J, K)
    lab = np.arange(nC, dtype=np.int64)
    own, nbr, lf, coarse, dirs = [], [], [], [], []
    for d, (dx, dy, dz) in enumerate(_DIRS):
        ni, nj, nk = I + dx, J + dy, K + dz
        same = idx.find(lvl, ni, nj, nk)
        if dx + dy + dz > 0:                       # same-level faces once, from the low side
            ok = same >= 0
            own.append(lab[ok]); nbr.append(same[ok]); lf.append(lvl[ok]); coarse.append(np.zeros(ok.sum(), dtype=bool))
            dirs.append(np.full(ok.sum(), d // 2, dtype=np.int8))
        miss = (same < 0) & (lvl > 0)
        par = idx.find(lvl[miss] - 1, ni[miss] >> 1, nj[miss] >> 1, nk[miss] >> 1)
        ok = par >= 0
        own.append(lab[miss][ok]); nbr.append(par[ok]); lf.append(lvl[miss][ok]); coarse.append(np.ones(ok.sum(), dtype=bool))
        dirs.append(np.full(ok.sum(), d // 2, dtype=np.int8))
    a, b = np.concatenate(own), np.concatenate(nbr)
    lf, coarse, dirs = np.concatenate(lf), np.concatenate(coarse), np.concatenate(dirs)
    l, u = np.minimum(a, b), np.maximum(a, b)
    order = np.argsort(l * np.int64(nC) + u, kind="stable")
    l, u, lf, coarse, dirs = l[order], u[order], lf[order], coarse[order], dirs[order]
    hf = m["h0"] / (1 << lf.astype(np.int64))
    area = hf * hf
    dn = np.where(coarse, 1.5 * hf, hf)            # h_fine/2 + h_coarse/2
    return l.astype(np.int32), u.astype(np.int32), area, dn, dirs


def problem(m=None, seed=12345, **kw):
    """The p-equation twin on the octree: laplacian coefficient |Sf|/(n.d) per face times the box stand-in's random
    factor (1 + 0.5 u01(seed, f)), diag = negSumDiag + fixedValue outlet (x = max) faces, the walls and
    the body zeroGradient as in motorBike/0.org/p; b = A x*, x* smooth in space; faceWeights = what
    faceAreaPairGAMGAgglomeration computes from the face area vectors (direction factors 1 / 1.01 / 1.02)."""
    if m is None:
        m = generate(**kw)
    l, u, area, dn, dirs = faces(m)
    # snappyHexMesh keeps the region reachable from locationInMesh (snappyHexMeshDict:159-162): pockets of cells cut off
    # by the castellation (all their neighbours inside the body) are dropped, the labels compacted in order
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    nC = m["level"].size
    ncomp, comp = connected_components(coo_matrix((np.ones(l.size, dtype=np.int8), (l, u)), shape=(nC, nC)), directed=False)
    if ncomp > 1:
        keep = comp == np.argmax(np.bincount(comp))
        new = np.cumsum(keep) - 1
        fk = keep[l] & keep[u]
        l, u = new[l[fk]].astype(np.int32), new[u[fk]].astype(np.int32)
        area, dn, dirs = area[fk], dn[fk], dirs[fk]
        m = dict(m, level=m["level"][keep], i=m["i"][keep], j=m["j"][keep], k=m["k"][keep])
    nC, nF = m["level"].size, l.size
    hmin = m["h0"] / (1 << m["maxLevel"])
    upper = -(1.0 + 0.5 * _cases.u01(seed, nF)) * (area / dn) / hmin
    diag = _cases._neg_sum_diag(nC, l, u, upper, upper)
    h = m["h0"] / (1 << m["level"].astype(np.int64))
    outlet = (m["i"].astype(np.int64) + 1) == (np.int64(m["base"][0]) << m["level"].astype(np.int64))
    diag[outlet] += (h[outlet] * h[outlet] / (0.5 * h[outlet])) / hmin
    x = m["lo"][0] + (m["i"] + 0.5) * h
    y = m["lo"][1] + (m["j"] + 0.5) * h
    z = m["lo"][2] + (m["k"] + 0.5) * h
    # faceAreaPairGAMGAgglomeration.C:59-72: mag(cmptMultiply(Sf/sqrt(magSf), vector(1, 1.01, 1.02))), Sf = area * e_dir
    comp = (area / np.sqrt(area)) * np.array([1.0, 1.01, 1.02])[dirs]
    p = dict(nCells=nC, lowerAddr=l, upperAddr=u, upper=upper, diag=diag, faceWeights=np.sqrt(comp * comp))
    xstar = np.sin(0.9 * x) * np.cos(1.7 * y) + 0.3 * np.sin(2.3 * z + 0.5 * x)
    p["source"] = _cases.amul(p, xstar)
    p["psi"] = np.zeros(nC)
    p["cellLevel"] = m["level"]
    return p
