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


def concatenate(subs):
    """Sub-domain mode (include/ldugpu.h: ldu_addr_set_subdomains): the K sub-domains of decompose() as ONE problem - cells of
    sub-domain 0, 1, ... one behind the other, the internal faces of each with its cell offset (still upper-triangular
    order), every processor patch (a -> b) as a cyclic patch paired with (b -> a).  Arithmetically the K-rank run of the
    reference (the reference itself emulates its ranks that way where it has no MPI: tests/test_fv_oracle_golden.py).
    Returns the problem dict with `subdomains` (sub-domain of every cell) and `cellStart`."""
    K = len(subs)
    start = np.concatenate(([0], np.cumsum([sp["nCells"] for sp in subs]))).astype(np.int64)
    pstart = np.concatenate(([0], np.cumsum([len(sp["patches"]) for sp in subs]))).astype(np.int64)
    out = dict(nCells=int(start[-1]),
               lowerAddr=np.concatenate([sp["lowerAddr"].astype(np.int64) + start[d] for d, sp in enumerate(subs)]).astype(np.int32),
               upperAddr=np.concatenate([sp["upperAddr"].astype(np.int64) + start[d] for d, sp in enumerate(subs)]).astype(np.int32))
    for k in ("diag", "upper", "lower", "source", "psi", "faceWeights"):
        if k in subs[0]:
            out[k] = np.concatenate([sp[k] for sp in subs])
    patches = []
    for d, sp in enumerate(subs):
        for q in sp["patches"]:
            patches.append(dict(faceCells=(q["faceCells"].astype(np.int64) + start[d]).astype(np.int32), bouCoeffs=q["bouCoeffs"],
                                intCoeffs=q["intCoeffs"], nbrDom=0, nbrRank=-1, nbrPatch=int(pstart[q["nbrDom"]] + q["nbrPatch"]),
                                cyclic=True))
    out["patches"] = patches
    out["patches_dev"] = [dict(faceCells=q["faceCells"], nbrRank=-1, nbrPatch=q["nbrPatch"], cyclic=True) for q in patches]
    out["subdomains"] = np.repeat(np.arange(K, dtype=np.int32), [sp["nCells"] for sp in subs])
    out["cellStart"] = start
    return out


def concatenated(p, cell_rank, n_ranks):
    """concatenate(decompose(p, cell_rank, n_ranks)[0]) without the per-rank passes (vectorised: the 12.7 M-cell mesh 64-way in
    seconds) -> (problem in sub-domain mode, global cell ids in the new order)"""
    cell_rank = np.asarray(cell_rank, dtype=np.int64)
    l, u = np.asarray(p["lowerAddr"], dtype=np.int64), np.asarray(p["upperAddr"], dtype=np.int64)
    upper = np.asarray(p["upper"])
    lower = np.asarray(p["lower"]) if "lower" in p else upper
    order = np.argsort(cell_rank, kind="stable")
    newid = np.empty(p["nCells"], dtype=np.int64)
    newid[order] = np.arange(p["nCells"])
    counts = np.bincount(cell_rank, minlength=n_ranks)
    rl, ru = cell_rank[l], cell_rank[u]
    inner = np.nonzero(rl == ru)[0]
    inner = inner[np.argsort(rl[inner], kind="stable")]
    out = dict(nCells=int(p["nCells"]), lowerAddr=newid[l[inner]].astype(np.int32), upperAddr=newid[u[inner]].astype(np.int32),
               diag=np.asarray(p["diag"])[order].copy(), upper=upper[inner].copy())
    if "lower" in p:
        out["lower"] = lower[inner].copy()
    for k in ("source", "psi"):
        if k in p:
            out[k] = np.asarray(p[k])[order].copy()
    if "faceWeights" in p:
        out["faceWeights"] = np.asarray(p["faceWeights"])[inner].copy()
    cut = np.nonzero(rl != ru)[0]
    own = np.concatenate([rl[cut], ru[cut]])
    nbr = np.concatenate([ru[cut], rl[cut]])
    face = np.concatenate([cut, cut])
    fc = np.concatenate([newid[l[cut]], newid[u[cut]]])
    bou = np.concatenate([-upper[cut], -lower[cut]])
    intc = np.concatenate([-lower[cut], -upper[cut]])
    o = np.lexsort((face, nbr, own))
    own, nbr, fc, bou, intc = own[o], nbr[o], fc[o], bou[o], intc[o]
    key = own * n_ranks + nbr
    starts = np.concatenate(([0], np.nonzero(np.diff(key))[0] + 1, [key.size])) if key.size else np.array([0])
    index = {}
    for i in range(len(starts) - 1):
        index[(int(own[starts[i]]), int(nbr[starts[i]]))] = i
    patches = []
    for i in range(len(starts) - 1):
        a, b = int(own[starts[i]]), int(nbr[starts[i]])
        sl = slice(starts[i], starts[i + 1])
        patches.append(dict(faceCells=fc[sl].astype(np.int32), bouCoeffs=bou[sl].copy(), intCoeffs=intc[sl].copy(), nbrDom=0,
                            nbrRank=-1, nbrPatch=index[(b, a)], cyclic=True))
    out["patches"] = patches
    out["patches_dev"] = [dict(faceCells=q["faceCells"], nbrRank=-1, nbrPatch=q["nbrPatch"], cyclic=True) for q in patches]
    out["subdomains"] = np.repeat(np.arange(n_ranks, dtype=np.int32), counts)
    out["cellStart"] = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    return out, order


def blob_ranks(nCells, lowerAddr, upperAddr, n_ranks):
    """n_ranks compact sub-domains of (nearly) equal size for any numbering (capi.partition_blobs: breadth-first blobs)"""
    from . import capi
    return capi.partition_blobs(nCells, lowerAddr, upperAddr, n_ranks).astype(np.int64)
