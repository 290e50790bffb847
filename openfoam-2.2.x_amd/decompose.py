"""Domain decomposition of an lduMatrix problem into per-rank sub-domains with processor
patches - what decomposePar produces for the reference: cells of a rank are
contiguous and locally numbered, inter-rank faces become `processor` patch faces with
faceCells addressing and interfaceBouCoeffs / interfaceIntCoeffs (not in upper/lower).

Sign convention (lduMatrixATmul.C:34-92 + processorFvPatchScalarField.C:125-128):
Amul does   result[faceCells[i]] -= bouCoeffs[i]*psiNbr[i]
so for a cut face f between owner P (rank a) and neighbour N (rank b):
    rank a:  bouCoeffs = -upper[f]   intCoeffs = -lower[f]     (row P multiplies psi_N by upper[f])
    rank b:  bouCoeffs = -lower[f]   intCoeffs = -upper[f]     (row N multiplies psi_P by lower[f])
"""
import numpy as np


def slab_ranks(nx, ny, nz, n_ranks):
    """Rank of each cell for a 1-D slab decomposition along z (k) of a natural-ordered box."""
    k = np.arange(nx * ny * nz) // (nx * ny)
    bounds = [(nz * r) // n_ranks for r in range(n_ranks + 1)]
    return np.searchsorted(bounds, k, side="right") - 1


def block_ranks(nx, ny, nz, px, py, pz):
    """Rank of each cell for a px*py*pz block decomposition (2x2x2)."""
    c = np.arange(nx * ny * nz)
    i, j, k = c % nx, (c // nx) % ny, c // (nx * ny)
    bi = np.minimum(i * px // nx, px - 1)
    bj = np.minimum(j * py // ny, py - 1)
    bk = np.minimum(k * pz // nz, pz - 1)
    return (bi + px * (bj + py * bk)).astype(np.int64)


def decompose(p, cell_rank, n_ranks, only_rank=None):
    """Split problem dict p (only_rank: build just that rank's sub-domain; the others are None).  Returns (subs, cell_maps): subs[r] is a problem dict with
    'patches' (oracle form: faceCells/bouCoeffs/intCoeffs/nbrDom/nbrPatch) and
    'patches_dev' (device form: faceCells/nbrRank); cell_maps[r] = global cell ids of rank r
    in local order (ascending global id, preserving the upper-triangular face order)."""
    l, u = np.asarray(p["lowerAddr"]), np.asarray(p["upperAddr"])
    upper = np.asarray(p["upper"])
    lower = np.asarray(p["lower"]) if "lower" in p else upper
    rl, ru = cell_rank[l], cell_rank[u]
    nC = p["nCells"]
    local_id = np.zeros(nC, dtype=np.int64)
    cell_maps = []
    for r in range(n_ranks):
        ids = np.nonzero(cell_rank == r)[0]
        local_id[ids] = np.arange(ids.size)
        cell_maps.append(ids)
    subs = []
    cut = rl != ru
    for r in range(n_ranks):
        if only_rank is not None and r != only_rank:
            subs.append(None)
            continue
        ids = cell_maps[r]
        inner = (rl == r) & (ru == r)
        sp = dict(nCells=int(ids.size), lowerAddr=local_id[l[inner]].astype(np.int32),
                  upperAddr=local_id[u[inner]].astype(np.int32), diag=np.asarray(p["diag"])[ids].copy(),
                  upper=upper[inner].copy())
        if "lower" in p:
            sp["lower"] = lower[inner].copy()
        for k in ("source", "psi"):
            if k in p:
                sp[k] = np.asarray(p[k])[ids].copy()
        if "faceWeights" in p:
            sp["faceWeights"] = np.asarray(p["faceWeights"])[inner].copy()
        # processor patches in ascending neighbour-rank order, faces in ascending global face id
        patches = []
        mine = cut & ((rl == r) | (ru == r))
        other = np.where(rl == r, ru, rl)
        for nb in sorted(set(other[mine].tolist())):
            fsel = np.nonzero(mine & (other == nb))[0]
            own_side = rl[fsel] == r
            fc = np.where(own_side, local_id[l[fsel]], local_id[u[fsel]]).astype(np.int32)
            bou = np.where(own_side, -upper[fsel], -lower[fsel])
            intc = np.where(own_side, -lower[fsel], -upper[fsel])
            patches.append(dict(faceCells=fc, bouCoeffs=bou, intCoeffs=intc, nbrDom=int(nb),
                                nbrRank=int(nb), faces=fsel))
        sp["patches"] = patches
        subs.append(sp)
    # pair patches: patch (r -> nb) matches patch (nb -> r); both list the same global faces
    for r in range(n_ranks):
        if subs[r] is None:
            continue
        if only_rank is not None:
            subs[r]["patches_dev"] = [dict(faceCells=q["faceCells"], nbrRank=q["nbrRank"])
                                      for q in subs[r]["patches"]]
            continue
        for q in subs[r]["patches"]:
            nb = q["nbrDom"]
            for j, q2 in enumerate(subs[nb]["patches"]):
                if q2["nbrDom"] == r:
                    assert np.array_equal(q["faces"], q2["faces"])
                    q["nbrPatch"] = j
        subs[r]["patches_dev"] = [dict(faceCells=q["faceCells"], nbrRank=q["nbrRank"])
                                  for q in subs[r]["patches"]]
    return subs, cell_maps


def gather(cell_maps, parts, nC):
    out = np.zeros(nC)
    for ids, x in zip(cell_maps, parts):
        out[ids] = x
    return out
