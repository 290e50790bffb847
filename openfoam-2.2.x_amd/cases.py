"""Synthetic lduMatrix problems as plain numpy arrays.

Every problem is a dict with the arrays OpenFOAM hands over at the
lduMatrix::solver boundary:
  nCells (int), lowerAddr/upperAddr int32[nF] (owner / neighbour, upper-triangular
  order: sorted by owner, then neighbour), diag f64[nC], upper f64[nF],
  lower f64[nF] (asymmetric only), source f64[nC], psi f64[nC] (initial guess),
  faceWeights f64[nF] (what faceAreaPairGAMGAgglomeration derives from Sf).

Random numbers come from a counter-based splitmix64 so that any language can
regenerate the identical matrix from (seed, face index).
"""
import numpy as np

_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def u01(seed, n):
    """n uniform doubles in [0,1): u01(seed, f) = (splitmix64(seed*2^32 + f) >> 11) * 2^-53."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        z = _splitmix64(idx)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def amul(p, x):
    """y = A x with the LDU face loop (numpy; order of summation NOT the reference's)."""
    l, u = p["lowerAddr"], p["upperAddr"]
    lower = p.get("lower", p["upper"])
    y = p["diag"] * x
    y += np.bincount(u, weights=lower * x[l], minlength=p["nCells"])
    y += np.bincount(l, weights=p["upper"] * x[u], minlength=p["nCells"])
    return y


def box_addressing(nx, ny, nz):
    """Owner/neighbour of a structured hex box in blockMesh's natural ordering (i fastest).

    Faces sorted by owner then neighbour (c+1 < c+nx < c+nx*ny), i.e. upper-triangular
    order (reference: lduAddressing.H:36-63).  Returns lowerAddr, upperAddr, dir (0/1/2).
    """
    nC = nx * ny * nz
    c = np.arange(nC, dtype=np.int64)
    i = c % nx
    j = (c // nx) % ny
    k = c // (nx * ny)
    valid = np.stack([i < nx - 1, j < ny - 1, k < nz - 1], axis=1)
    nbr = np.stack([c + 1, c + nx, c + nx * ny], axis=1)
    own = np.repeat(c[:, None], 3, axis=1)
    d = np.tile(np.arange(3, dtype=np.int8), (nC, 1))
    m = valid.ravel()
    return (own.ravel()[m].astype(np.int32), nbr.ravel()[m].astype(np.int32), d.ravel()[m])


def _neg_sum_diag(nC, l, u, lower, upper):
    """lduMatrix::negSumDiag (lduMatrixOperations.C:50-64): diag[l] -= lower; diag[u] -= upper."""
    d = np.zeros(nC)
    d -= np.bincount(l, weights=lower, minlength=nC)
    d -= np.bincount(u, weights=upper, minlength=nC)
    return d


def laplacian2d(nx=40, ny=40):
    """SURVEY.md 8c known-answer case: 5-point Laplacian, upper=-1, diag=-sum(offdiag),
    diag[0]+=1, b_i=sin(0.37 i), psi0=0.  Reference: DICPCG 78 iterations to 9.4088e-11."""
    l, u, d = box_addressing(nx, ny, 1)
    nC = nx * ny
    upper = -np.ones(l.size)
    diag = _neg_sum_diag(nC, l, u, upper, upper)
    diag[0] += 1.0
    return dict(nCells=nC, lowerAddr=l, upperAddr=u, diag=diag, upper=upper,
                source=np.sin(0.37 * np.arange(nC)), psi=np.zeros(nC),
                faceWeights=np.where(d == 0, 1.0, 1.01))


def box3d(nx, ny=None, nz=None, asym=False, seed=12345, seed_asym=54321):
    """SURVEY.md 8d C3 stand-in for the 10 M-cell motorBike p-matrix (n=216):
    variable-coefficient 7-point Laplacian upper[f] = -(1 + 0.5*u01(seed,f))*a_dir,
    a=(1,1.01,1.02); symmetric: diag=-sum(offdiag); asym: lower = upper - 0.3*(2*u01(seed2,f)-1),
    diag = negSumDiag.  One reference row diag[0]*=2, b = A x*, x*_i = sin(1e-3 i), psi0 = 0."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    l, u, d = box_addressing(nx, ny, nz)
    nC = nx * ny * nz
    nF = l.size
    a = np.array([1.0, 1.01, 1.02])[d]
    upper = -(1.0 + 0.5 * u01(seed, nF)) * a
    p = dict(nCells=nC, lowerAddr=l, upperAddr=u, upper=upper, faceWeights=a.copy())
    if asym:
        phi = 0.3 * (2.0 * u01(seed_asym, nF) - 1.0)
        lower = upper - phi
        p["lower"] = lower
        diag = _neg_sum_diag(nC, l, u, lower, upper)
    else:
        diag = _neg_sum_diag(nC, l, u, upper, upper)
    diag[0] *= 2.0
    p["diag"] = diag
    xstar = np.sin(1e-3 * np.arange(nC))
    p["source"] = amul(p, xstar)
    p["psi"] = np.zeros(nC)
    return p


def asymmetric(p, seed=54321, strength=0.3):
    """SURVEY.md 8d's recipe for config C3's other half (the U-equation solvers) on ANY symmetric problem's addressing:
    lower[f] = upper[f] - phi[f] with a face flux phi[f] = strength * (2 u01(seed, f) - 1) * |upper[f]| (on the box, where
    |upper| ~ 1, this is box3d(asym=True)'s operator family), the diagonal re-summed the way negSumDiag does
    (lduMatrixOperations.C:50-64: diag[l] -= lower, diag[u] -= upper), so what the symmetric problem added to its diagonal
    (a fixedValue patch, a reference row) stays; b = A x* for the smooth x* the symmetric problem was made with, where
    known, else sin(1e-3 i)."""
    l, u, nC = p["lowerAddr"], p["upperAddr"], p["nCells"]
    up = p["upper"]
    phi = strength * (2.0 * u01(seed, up.size) - 1.0) * np.abs(up)
    lower = up - phi
    q = dict(p)
    q["lower"] = lower
    # negSumDiag of the symmetric matrix took `upper` on both sides: the l side now takes `lower`
    q["diag"] = p["diag"] + np.bincount(l, weights=phi, minlength=nC)
    q["source"] = amul(q, np.sin(1e-3 * np.arange(nC)))
    q["psi"] = np.zeros(nC)
    return q


def _u01_key(seed, key):
    """uniform double in [0,1) per 64-bit key (a global cell id * 3 + face direction): the same on every rank"""
    with np.errstate(over="ignore"):
        z = _splitmix64(key.astype(np.uint64) + (np.uint64(seed) << np.uint64(44)))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def box3d_block(n, blocks, rank, seed=12345):
    """One rank's sub-domain of a (px n) x (py n) x (pz n) hex box cut into px x py x pz blocks of n^3 cells (weak scaling:
    every rank holds n^3 cells whatever the rank count), generated WITHOUT the global matrix: the coefficient of a face is a
    function of (global owner cell, direction), so both ranks of a cut face compute the same value.  Same operator family as
    box3d: upper = -(1 + 0.5 u01) a_dir, diag = -sum(offdiag) over internal AND cut faces, diag of global cell 0 doubled,
    b = A x*, x*_g = sin(1e-3 g).  Patches as decomposePar orders them (ascending neighbour rank, faces in ascending global
    face order = ascending boundary cell): device form `patches_dev` and the coefficient arrays in `patches`."""
    px, py, pz = blocks
    NX, NY = px * n, py * n
    bi, bj, bk = rank % px, (rank // px) % py, rank // (px * py)
    l, u, d = box_addressing(n, n, n)
    nC = n ** 3
    c = np.arange(nC, dtype=np.int64)
    i, j, k = c % n, (c // n) % n, c // (n * n)
    g = (bi * n + i) + NX * ((bj * n + j) + NY * (bk * n + k))           # global cell ids of the local cells
    a_dir = np.array([1.0, 1.01, 1.02])
    stride = np.array([1, NX, NX * NY], dtype=np.int64)

    def coeff(owner_g, dirn):
        return -(1.0 + 0.5 * _u01_key(seed, owner_g * 3 + dirn)) * a_dir[dirn]

    upper = coeff(g[l], d.astype(np.int64))
    diag = _neg_sum_diag(nC, l, u, upper, upper)
    xs = np.sin(1e-3 * g)
    src_cut = np.zeros(nC)
    patches = []
    # ascending neighbour rank: -z, -y, -x, +x, +y, +z
    for dirn, sign in ((2, -1), (1, -1), (0, -1), (0, 1), (1, 1), (2, 1)):
        b = (bi, bj, bk)[dirn] + sign
        if b < 0 or b >= blocks[dirn]:
            continue
        idx = (i, j, k)[dirn]
        cells = np.nonzero(idx == (0 if sign < 0 else n - 1))[0]
        gn = g[cells] + sign * stride[dirn]                               # the cell across the cut
        owner_g = np.where(sign > 0, g[cells], gn)
        cf = coeff(owner_g, np.full(cells.size, dirn, dtype=np.int64))
        nb = rank + sign * (1, px, px * py)[dirn]
        np.subtract.at(diag, cells, cf)
        np.add.at(src_cut, cells, cf * np.sin(1e-3 * gn))
        patches.append(dict(faceCells=cells.astype(np.int32), bouCoeffs=-cf, intCoeffs=-cf, nbrDom=int(nb), nbrRank=int(nb)))
    if rank == 0:
        diag[0] *= 2.0
    p = dict(nCells=nC, lowerAddr=l, upperAddr=u, upper=upper, diag=diag, faceWeights=a_dir[d].copy(), psi=np.zeros(nC))
    p["source"] = amul(p, xs) + src_cut
    p["patches"] = patches
    p["patches_dev"] = [dict(faceCells=q["faceCells"], nbrRank=q["nbrRank"]) for q in patches]
    return p


def jump2d(nx, ny, ratio=1000.0):
    """SURVEY.md 8d C5 twin (damBreak p_rgh): 2-D 5-point with the face coefficient jumping
    1 <-> ratio across the diagonal i+j = (nx+ny)/2 (two-phase density ratio)."""
    l, u, d = box_addressing(nx, ny, 1)
    nC = nx * ny
    il, jl = l % nx, l // nx
    iu, ju = u % nx, u // nx
    heavy = ((il + jl) + (iu + ju)) < (nx + ny)
    upper = -np.where(heavy, ratio, 1.0)
    diag = _neg_sum_diag(nC, l, u, upper, upper)
    diag[0] *= 2.0
    p = dict(nCells=nC, lowerAddr=l, upperAddr=u, diag=diag, upper=upper,
             faceWeights=np.where(d == 0, 1.0, 1.01), psi=np.zeros(nC))
    p["source"] = amul(p, np.cos(3e-3 * np.arange(nC)))
    return p


def random_graph(nC, avg_deg=4, band=50, asym=False, seed=7):
    """Irregular ('unstructured') addressing: each cell connects to a few random
    higher-numbered cells within a band.  Exercises ragged rows, empty rows and
    non-uniform dependency levels in the triangular sweeps."""
    rng = np.random.RandomState(seed)
    own = []
    nbr = []
    for c in range(nC - 1):
        k = rng.poisson(avg_deg / 2.0)
        if k == 0:
            continue
        hi = min(nC - 1, c + band)
        cand = np.unique(rng.randint(c + 1, hi + 1, size=k))
        own.extend([c] * cand.size)
        nbr.extend(cand.tolist())
    l = np.array(own, dtype=np.int32)
    u = np.array(nbr, dtype=np.int32)
    nF = l.size
    upper = -(0.5 + rng.rand(nF))
    p = dict(nCells=nC, lowerAddr=l, upperAddr=u, upper=upper,
             faceWeights=0.5 + rng.rand(nF))
    if asym:
        lower = upper - 0.3 * (2 * rng.rand(nF) - 1)
        p["lower"] = lower
        diag = _neg_sum_diag(nC, l, u, lower, upper)
    else:
        diag = _neg_sum_diag(nC, l, u, upper, upper)
    diag += 0.05 + 0.1 * rng.rand(nC)
    p["diag"] = diag
    p["source"] = rng.randn(nC)
    p["psi"] = 0.1 * rng.randn(nC)
    return p


def random_graph_fast(nC, avg_deg=7.0, band=600, seed=11):
    """Large irregular addressing, vectorised (random_graph above is a Python loop): cell c draws Poisson(avg_deg/2)
    higher-numbered neighbours within `band`; duplicates are dropped; faces come out in upper-triangular order.
    Symmetric M-matrix coefficients, b = A x* with x*_i = sin(1e-3 i).  The stand-in for an unstructured (snappyHexMesh-
    like) matrix in bench.py --mesh random: 5-9 neighbours per cell for avg_deg 7."""
    rng = np.random.RandomState(seed)
    k = rng.poisson(avg_deg / 2.0, size=nC).astype(np.int64)
    k[k > band] = band
    own = np.repeat(np.arange(nC, dtype=np.int64), k)
    nbr = own + rng.randint(1, band + 1, size=own.size)
    keep = nbr < nC
    key = np.unique(own[keep] * np.int64(nC) + nbr[keep])     # sorted by (owner, neighbour), duplicates removed
    l = (key // nC).astype(np.int32)
    u = (key % nC).astype(np.int32)
    nF = l.size
    upper = -(0.5 + u01(seed, nF))
    diag = _neg_sum_diag(nC, l, u, upper, upper)
    diag += 0.05
    p = dict(nCells=nC, lowerAddr=l, upperAddr=u, upper=upper, diag=diag, faceWeights=0.5 + u01(seed + 1, nF))
    p["source"] = amul(p, np.sin(1e-3 * np.arange(nC)))
    p["psi"] = np.zeros(nC)
    return p


def irregular_box(n, extra=0.6, seed=21):
    """Unstructured-like 3-D addressing with the locality of a real mesh: the n^3 hex box plus, per cell and with
    probability `extra` each, a face to one of its diagonal neighbours (+x+y), (+y+z), (+x+z) - what split-hex /
    polyhedral cells of a snappyHexMesh refinement region look like to the matrix: 6 ... 12 neighbours per cell
    (8 on average for extra = 0.6... ), ragged rows, no i+j+k wavefront structure for the sweeps to lean on.
    Symmetric M-matrix, b = A x*.  bench.py --mesh irregular renumbers it with Foam::bandCompression first."""
    nC = n ** 3
    c = np.arange(nC, dtype=np.int64)
    i, j, k = c % n, (c // n) % n, c // (n * n)
    rng = np.random.RandomState(seed)
    own, nbr = [], []
    for (di, dj, dk), prob in (((1, 0, 0), 1.0), ((0, 1, 0), 1.0), ((0, 0, 1), 1.0),
                               ((1, 1, 0), extra), ((0, 1, 1), extra), ((1, 0, 1), extra)):
        ok = (i + di < n) & (j + dj < n) & (k + dk < n)
        if prob < 1.0:
            ok &= rng.rand(nC) < prob
        own.append(c[ok])
        nbr.append(c[ok] + di + dj * n + dk * n * n)
    own, nbr = np.concatenate(own), np.concatenate(nbr)
    key = np.sort(own * np.int64(nC) + nbr)
    l, u = (key // nC).astype(np.int32), (key % nC).astype(np.int32)
    nF = l.size
    upper = -(0.5 + u01(seed, nF))
    diag = _neg_sum_diag(nC, l, u, upper, upper)
    diag[0] *= 2.0
    p = dict(nCells=nC, lowerAddr=l, upperAddr=u, upper=upper, diag=diag, faceWeights=0.5 + u01(seed + 1, nF))
    p["source"] = amul(p, np.sin(1e-3 * np.arange(nC)))
    p["psi"] = np.zeros(nC)
    return p


def renumbered(p, order, faceMap, flip, newLower, newUpper):
    """the same matrix after a cell renumbering (capi.renumber_addressing): cell order[i] becomes cell i"""
    q = dict(nCells=p["nCells"], lowerAddr=newLower, upperAddr=newUpper, diag=p["diag"][order],
             source=p["source"][order], psi=p["psi"][order])
    up = p["upper"][faceMap]
    if "lower" in p:
        lo = p["lower"][faceMap]
        q["upper"] = np.where(flip == 1, lo, up)
        q["lower"] = np.where(flip == 1, up, lo)
    else:
        q["upper"] = up
    if "faceWeights" in p:
        q["faceWeights"] = p["faceWeights"][faceMap]
    return q


def to_ldub_dict(p):
    out = {"nCells": np.array([p["nCells"]], dtype=np.int32)}
    for k in ("lowerAddr", "upperAddr", "diag", "upper", "lower", "source", "psi", "faceWeights"):
        if k in p:
            out[k] = p[k]
    return out
