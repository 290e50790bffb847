"""TOOL (CPU only): how the cell numbering - which the reference lets the user choose (renumberMesh,
src/renumber/renumberMethods/manualRenumber/manualRenumber.C) - shapes what the sweeps of a GAMG solve cost here: the depth
of the GaussSeidel dependency DAG of EVERY level of the hierarchy (the coarse numberings follow from the fine one through the
sequential pair matching, pairGAMGAgglomerate.C:31-198, and its reversal), the steps of k pipelined sweeps, the gather
locality, and the number of V-cycles a solve needs (CPU oracle).

    python tools/numbering_probe.py MESH ORDERING [ORDERING ...] [--solve] [--export DIR]

MESH = mbtut | mb2 | mb12 (data/motorbike).  ORDERING:
    snappy          the numbering snappyHexMesh / hexRef8 left
    rcm             Foam::bandCompression (what renumberMesh applies by default; the bench's numbering so far)
    cm              the same, not reversed
    rand[:seed]     a random permutation
    mc[:within]     multi-colour: greedy colouring (cells taken in rcm order), colours numbered one after the other;
                    inside a colour: within = rcm | rand | tileK (tiles of K consecutive rcm positions in random order)
    shell[:within]  breadth-first shells of the rcm traversal, inside a shell an independent set first (colour, then within)
    blob:K[:within] K breadth-first blobs one after the other, multi-colour inside a blob
    hmc[:iters]     hierarchical multi-colour (see hmc_order)
    tilerand:K      tiles of K consecutive rcm positions in random order (numpy RNG), rcm order inside a tile
    tiles:K[:seed]  the same by the library's deterministic ldu_tile_shuffle (what a manualRenumber file would hold)
    sshell:S[:w]    super-shells of S breadth-first shells, multi-colour inside
The table: per level cells / dependency levels of one sweep / steps of 2 and 4 pipelined sweeps; their sums over the levels;
distinct 128-byte lines per 64-row gather on the finest level; --solve: V-cycles and residual history of the bench's solve."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
from openfoam_amd import capi, cases, motorbike  # noqa: E402

GAMG = dict(solver="GAMG", tolerance=1e-7, relTol=0.01, smoother="GaussSeidel", nPreSweeps=0, nPostSweeps=2, nFinestSweeps=2,
            cacheAgglomeration=1, agglomerator="faceAreaPair", nCellsInCoarsestLevel=10, mergeLevels=1)


def helper():
    so = os.path.join("/tmp", "libdag_depth.so")
    src = os.path.join(ROOT, "tools", "dag_depth.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
    L = C.CDLL(so)
    L.gather_lines.restype = C.c_double
    return L


H = helper()


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def dag_levels(nC, l, u):
    l, u = _i(l), _i(u)
    lev = np.zeros(nC, dtype=np.int32)
    return H.dag_levels(int(nC), int(l.size), _p(l), _p(u), _p(lev)), lev


def dag_steps(nC, l, u, k):
    l, u = _i(l), _i(u)
    return H.dag_steps_k(int(nC), int(l.size), _p(l), _p(u), int(k), None)


def colour(nC, l, u, order):
    l, u, order = _i(l), _i(u), _i(order)
    col = np.zeros(nC, dtype=np.int32)
    n = H.greedy_colour(int(nC), int(l.size), _p(l), _p(u), _p(order), _p(col))
    return n, col


def gather_lines(nC, l, u, pos):
    l, u, pos = _i(l), _i(u), _i(pos)
    return H.gather_lines(int(nC), int(l.size), _p(l), _p(u), _p(pos))


def renumber(p, order):
    nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], _i(order))
    return cases.renumbered(p, _i(order), fmap, flip, nl, nu)


def within_key(kind, nC, rcm_pos, rng):
    """secondary key of a cell inside its colour / shell / blob"""
    if kind == "rcm":
        return rcm_pos.astype(np.int64)
    if kind == "rand":
        return rng.permutation(nC).astype(np.int64)
    if kind.startswith("tile"):
        K = int(kind[4:])
        tile = rcm_pos // K
        tperm = rng.permutation(int(tile.max()) + 1)
        return tperm[tile].astype(np.int64) * K + (rcm_pos % K)
    raise SystemExit("unknown within-order " + kind)


def bfs_shells(nC, l, u, order):
    """shell index of every cell: breadth-first distance from the first cell of `order` (restarts at the next unvisited cell
    of `order` for disconnected parts)"""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as cg
    A = sp.coo_matrix((np.ones(l.size, dtype=np.int8), (l, u)), shape=(nC, nC)).tocsr()
    A = A + A.T
    shell = np.full(nC, -1, dtype=np.int64)
    base = 0
    for c in order:
        if shell[c] >= 0:
            continue
        d = cg.shortest_path(A, method="D", unweighted=True, indices=int(c))
        m = np.isfinite(d) & (shell < 0)
        shell[m] = base + d[m].astype(np.int64)
        base = int(shell.max()) + 1
        if (shell >= 0).all():
            break
    return shell


def hmc_order(p, rcm, iters, rng, verbose=True):
    """hierarchical multi-colour.  The coarse cells of a level are numbered in the REVERSED order in which the pair matching
    reaches their first cell (pairGAMGAgglomerate.C:83-197), i.e. by the fine numbering.  So: build the hierarchy under the
    current numbering, colour every level's graph, and give a fine cell the key (own colour, key of its coarse cell) where the
    key of a coarse cell is again (its colour, key of ITS coarse cell), most significant first - the coarse cells of a level
    are then created colour by colour (descending, because of the reversal), which makes the coarse numbering (nearly)
    multi-colour too.  The pairing itself depends on the numbering, so this is a fixed-point iteration; `iters` rounds."""
    import oracle_py
    nC = p["nCells"]
    order = mc_order(p, rcm, "rcm", rng)
    for it in range(iters):
        q = renumber(p, order)
        lv = oracle_py.System(q).gamg_levels(**GAMG)
        # key of the cells of each level, coarsest first
        nL = len(lv)
        keys = [None] * (nL + 1)       # keys[0] = finest
        ncs = [nC] + [L["nCells"] for L in lv]
        addr = [(q["lowerAddr"], q["upperAddr"])] + [(L["lowerAddr"], L["upperAddr"]) for L in lv]
        for li in range(nL, -1, -1):
            n = ncs[li]
            l_, u_ = addr[li]
            ncol, col = colour(n, l_, u_, np.arange(n, dtype=np.int32))
            if li == nL:
                parent_rank = np.zeros(n, dtype=np.int64)
            else:
                # rank wanted for the parent in ITS level's final numbering; the pair matching numbers coarse cells in visiting
                # order and then reverses: parent of rank r should be created r-th from the END, i.e. visited late when r is small
                par = lv[li]["restrict"].astype(np.int64)          # cell of level li -> cell of level li + 1
                pk = keys[li + 1]
                want = np.empty(ncs[li + 1], dtype=np.int64)
                want[np.argsort(pk, kind="stable")] = np.arange(ncs[li + 1])
                parent_rank = (ncs[li + 1] - 1 - want)[par]
            k = col.astype(np.int64) * (int(parent_rank.max()) + 1) + parent_rank
            # (ties: current position)
            keys[li] = k * n + np.arange(n)
        new_local = np.argsort(keys[0], kind="stable")             # positions of q in the new order
        order = _i(order)[new_local]
        if verbose:
            print("  hmc round %d: %d levels" % (it, nL), flush=True)
    return order


def mc_order(p, rcm, within, rng):
    nC = p["nCells"]
    rcm_pos = np.empty(nC, dtype=np.int64)
    rcm_pos[rcm] = np.arange(nC)
    ncol, col = colour(nC, p["lowerAddr"], p["upperAddr"], rcm)
    key2 = within_key(within, nC, rcm_pos, rng)
    return np.lexsort((key2, col)).astype(np.int32)


def make_order(p, spec, rcm, seed=1):
    nC = p["nCells"]
    rng = np.random.RandomState(seed)
    f = spec.split(":")
    rcm_pos = np.empty(nC, dtype=np.int64)
    rcm_pos[rcm] = np.arange(nC)
    if f[0] == "snappy":
        return np.arange(nC, dtype=np.int32)
    if f[0] == "rcm":
        return rcm
    if f[0] == "cm":
        return rcm[::-1].copy()
    if f[0] == "rand":
        return np.random.RandomState(int(f[1]) if len(f) > 1 else seed).permutation(nC).astype(np.int32)
    if f[0] == "mc":
        return mc_order(p, rcm, f[1] if len(f) > 1 else "rcm", rng)
    if f[0] == "shell":
        shell = bfs_shells(nC, p["lowerAddr"], p["upperAddr"], rcm)
        ncol, col = colour(nC, p["lowerAddr"], p["upperAddr"], rcm)
        key2 = within_key(f[1] if len(f) > 1 else "rcm", nC, rcm_pos, rng)
        return np.lexsort((key2, col, shell)).astype(np.int32)
    if f[0] == "sshell":
        # super-shells of S breadth-first shells one after the other (the sweep still runs through the domain front by front),
        # multi-colour inside a super-shell
        S = int(f[1])
        shell = bfs_shells(nC, p["lowerAddr"], p["upperAddr"], rcm) // S
        ncol, col = colour(nC, p["lowerAddr"], p["upperAddr"], rcm)
        key2 = within_key(f[2] if len(f) > 2 else "rcm", nC, rcm_pos, rng)
        return np.lexsort((key2, col, shell)).astype(np.int32)
    if f[0] == "tilerand":
        # tiles of K consecutive bandCompression positions in random order, bandCompression order inside a tile (no colours)
        return np.argsort(within_key("tile" + f[1], nC, rcm_pos, rng), kind="stable").astype(np.int32)
    if f[0] == "stride":
        # tiles of K consecutive bandCompression positions taken with stride S: tiles 0, S, 2S, ... then 1, S+1, ... (stride:K:S)
        K, S = int(f[1]), int(f[2])
        tile = rcm_pos // K
        return np.lexsort((rcm_pos, tile // S, tile % S)).astype(np.int32)
    if f[0] == "tiles":
        # the library's own: ldu_tile_shuffle(bandCompression order, K, seed) - tiles:K[:seed]
        return capi.tile_shuffle(rcm, int(f[1]), int(f[2]) if len(f) > 2 else 1)
    if f[0] == "blob":
        K = int(f[1])
        part = capi.partition_blobs(nC, p["lowerAddr"], p["upperAddr"], K).astype(np.int64)
        ncol, col = colour(nC, p["lowerAddr"], p["upperAddr"], rcm)
        key2 = within_key(f[2] if len(f) > 2 else "rcm", nC, rcm_pos, rng)
        return np.lexsort((key2, col, part)).astype(np.int32)
    if f[0] == "hmc":
        return hmc_order(p, rcm, int(f[1]) if len(f) > 1 else 2, rng)
    raise SystemExit("unknown ordering " + spec)


def evaluate(p, order, solve=False, lines=True):
    import oracle_py
    q = renumber(p, order)
    t0 = time.time()
    S = oracle_py.System(q)
    lv = S.gamg_levels(**GAMG)
    t_h = time.time() - t0
    rows = []
    addr = [(q["nCells"], q["lowerAddr"], q["upperAddr"])] + [(L["nCells"], L["lowerAddr"], L["upperAddr"]) for L in lv]
    for n, l_, u_ in addr:
        d1, _ = dag_levels(n, l_, u_)
        rows.append(dict(nCells=int(n), nFaces=int(l_.size), levels=int(d1), steps2=int(dag_steps(n, l_, u_, 2)),
                         steps4=int(dag_steps(n, l_, u_, 4))))
    out = dict(levels=rows, nLevels=len(rows), hierarchy_s=t_h,
               sum_levels=sum(r["levels"] for r in rows), sum_steps2=sum(r["steps2"] for r in rows),
               sum_steps4=sum(r["steps4"] for r in rows))
    if lines:
        # storage in dependency-level order (the engines' layout) and in the numbering itself
        n, l_, u_ = addr[0]
        _, lev = dag_levels(n, l_, u_)
        pos_level = np.empty(n, dtype=np.int32)
        pos_level[np.argsort(lev, kind="stable")] = np.arange(n, dtype=np.int32)
        out["lines_level_order"] = gather_lines(n, l_, u_, pos_level)
        out["lines_natural"] = gather_lines(n, l_, u_, np.arange(n, dtype=np.int32))
    if solve:
        t0 = time.time()
        x, perf = S.solve(q["psi"], q["source"], **GAMG)
        out["solve_s_oracle_1core"] = time.time() - t0
        out["vcycles"] = int(perf["nIterations"])
        out["history"] = [float(v) for v in perf["history"]]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mesh")
    ap.add_argument("orderings", nargs="+")
    ap.add_argument("--solve", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--export", default=None, help="directory for manualRenumber files (cellMap: new cell i <- old cell order[i])")
    args = ap.parse_args()
    p = motorbike.problem(args.mesh)
    p.pop("cellLevel"); p.pop("meta")
    nC = p["nCells"]
    t0 = time.time()
    rcm = capi.band_compression(nC, p["lowerAddr"], p["upperAddr"])
    print("%s: %d cells, %d faces; bandCompression %.1f s" % (args.mesh, nC, p["lowerAddr"].size, time.time() - t0), flush=True)
    res = {}
    for spec in args.orderings:
        t0 = time.time()
        order = make_order(p, spec, rcm)
        assert np.array_equal(np.sort(order), np.arange(nC))
        t_o = time.time() - t0
        r = evaluate(p, order, solve=args.solve)
        r["ordering_s"] = t_o
        res[spec] = r
        print("== %-14s levels %2d | one sweep: sum %5d (finest %4d) | 2 sweeps: sum %5d | 4 sweeps: sum %5d | lines/gather level-order "
              "%.1f natural %.1f%s" % (spec, r["nLevels"], r["sum_levels"], r["levels"][0]["levels"], r["sum_steps2"], r["sum_steps4"],
                                      r["lines_level_order"], r["lines_natural"],
                                      (" | V-cycles %d final %.3e" % (r["vcycles"], r["history"][-1])) if args.solve else ""), flush=True)
        print("   cells : " + " ".join("%d" % L["nCells"] for L in r["levels"]))
        print("   levels: " + " ".join("%d" % L["levels"] for L in r["levels"]))
        print("   steps2: " + " ".join("%d" % L["steps2"] for L in r["levels"]), flush=True)
        if args.export:
            os.makedirs(args.export, exist_ok=True)
            np.save(os.path.join(args.export, "%s_%s.npy" % (args.mesh, spec.replace(":", "_"))), order)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
