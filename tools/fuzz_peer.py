"""Randomised stress of the MULTI-RANK paths on the peer-store carrier (not a test: a hunt for rare failures): N processes
(one rank each, sharing the visible GPUs), random matrices cut into N sub-domains at random, every rank compares its slice of
Amul / Tmul / residual / GaussSeidel (1-4 sweeps) / symGaussSeidel with the multi-domain oracle bit for bit and the GAMG / Krylov
histories to 1e-6 - the halo exchange by peer stores, the rank-ordered all-reduce kernel, the distributed coarsest-level Krylov
solve in one kernel (coarsest_krylov_peer_kernel) and the one-launch smoother of small patched levels (gs_wg_peer_kernel).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/fuzz_peer.py [seconds] [seed]"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rank, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    entry.load_package()
    from openfoam_amd import capi, cases, decompose
    import oracle_py as O
    O.build()
    rng = np.random.RandomState(seed)          # the same stream on every rank
    ctx = capi.Context(rank % max(1, torch.cuda.device_count()))
    ctx.comm_init_peer(rank, n, capi.oob_torch())
    t0 = time.time()
    n_cases = n_checks = 0
    bad = []
    while True:
        # (every rank must leave the loop in the same iteration: rank 0's clock decides)
        go = [time.time() - t0 < budget]
        dist.broadcast_object_list(go, src=0)
        if not go[0] or bad:
            break
        kind = int(rng.randint(4))
        asym = bool(rng.randint(2))
        if kind == 0:
            dims = [int(rng.randint(2, 14)) for _ in range(3)]
            p = cases.box3d(*dims, asym=asym)
            nC = p["nCells"]
            mode = int(rng.randint(3))
            if mode == 0:       # blocks along the numbering
                cr = (np.arange(nC, dtype=np.int64) * n) // nC
            elif mode == 1:     # random assignment in chunks of a few cells: many small patches, every rank neighbours every rank
                ch = int(rng.randint(1, 9))
                cr = rng.randint(0, n, size=(nC + ch - 1) // ch).repeat(ch)[:nC].astype(np.int64)
            else:               # geometric slabs of unequal size
                cuts = np.sort(rng.choice(np.arange(1, nC), size=n - 1, replace=False)) if nC > n else np.arange(1, n)
                cr = np.searchsorted(cuts, np.arange(nC), side="right").astype(np.int64)
        elif kind == 1:
            nC = int(rng.randint(n * 4, 3000))
            p = cases.random_graph(nC, int(rng.randint(2, 9)), int(rng.randint(3, max(4, nC // 3))), asym=asym)
            cr = (np.arange(nC, dtype=np.int64) * n) // nC
        elif kind == 2:
            nC = int(rng.randint(2000, 30000))
            p = cases.random_graph(nC, int(rng.randint(3, 12)), int(rng.randint(20, 600)), asym=asym)
            ch = int(rng.randint(50, 2000))
            cr = rng.randint(0, n, size=(nC + ch - 1) // ch).repeat(ch)[:nC].astype(np.int64)
        else:
            m = int(rng.randint(12, 40))
            p = cases.box3d(m, m, int(rng.randint(2, 12)), asym=asym)
            nC = p["nCells"]
            cr = (np.arange(nC, dtype=np.int64) * n) // nC
        for r in range(n):      # no empty rank (decomposePar never writes one)
            if not np.any(cr == r):
                cr[r] = r
        if "faceWeights" not in p:
            p["faceWeights"] = 0.5 + rng.rand(p["lowerAddr"].size)
        subs, maps = decompose.decompose(p, cr, n)
        S = O.System(subs)
        sp = subs[rank]
        off = sum(s["nCells"] for s in subs[:rank])
        sl = slice(off, off + sp["nCells"])
        a = capi.Addressing(ctx, sp["nCells"], sp["lowerAddr"], sp["upperAddr"], sp.get("faceWeights"), patches=sp["patches_dev"])
        mt = capi.Matrix(a)
        mt.set_coeffs(sp["diag"], sp["upper"], sp.get("lower"))
        for i, q in enumerate(sp["patches"]):
            mt.set_patch_coeffs(i, q["bouCoeffs"], q["intCoeffs"])
        xs = [rng.randn(s["nCells"]) for s in subs]
        bs = [rng.randn(s["nCells"]) for s in subs]
        X, B = np.concatenate(xs), np.concatenate(bs)
        k = int(rng.randint(1, 5))
        checks = [("Amul", mt.Amul(xs[rank]), S.Amul(X)), ("Tmul", mt.Tmul(xs[rank]), S.Tmul(X)),
                  ("residual", mt.residual(xs[rank], bs[rank]), S.residual(X, B)),
                  ("GaussSeidel%d" % k, mt.smooth("GaussSeidel", xs[rank], bs[rank], k), S.smooth("GaussSeidel", X, B, k)),
                  ("symGaussSeidel", mt.smooth("symGaussSeidel", xs[rank], bs[rank], 1), S.smooth("symGaussSeidel", X, B, 1))]
        for name, got, want in checks:
            n_checks += 1
            if not np.array_equal(got, want[sl]):
                d = np.flatnonzero(got != want[sl])
                brows = set(int(c) for q in sp["patches"] for c in q["faceCells"])
                bad.append((n_cases, name, "%d of %d cells differ, %d of them boundary cells, max |diff| %.3e, first cells %s"
                            % (d.size, got.size, sum(int(c) in brows for c in d), float(np.abs(got - want[sl]).max()), d[:6].tolist())))
        anybad = [None] * n
        dist.all_gather_object(anybad, bool(bad))
        if any(anybad):
            # diagnosis (every rank takes part: the operations are collective): the same Tmul / Amul again
            t2, a2 = mt.Tmul(xs[rank]), mt.Amul(xs[rank])
            if bad:
                loc = dict(sp); loc["patches"] = []
                t0_ = O.System([loc]).Tmul(xs[rank])
                bad.append(("zero-intCoeffs candidate", "first Tmul == Tmul without interface terms: %s; repeat: %s" % (
                    np.array_equal(checks[1][1], t0_), np.array_equal(t2, t0_))))
                bad.append(("again", "Tmul ok now: %s, Amul ok now: %s, first Tmul == Amul result: %s" % (
                    np.array_equal(t2, S.Tmul(X)[sl]), np.array_equal(a2, S.Amul(X)[sl]), np.array_equal(checks[1][1], checks[0][1]))))
        X0 = np.concatenate([s["psi"] for s in subs]); B0 = np.concatenate([s["source"] for s in subs])
        ncl = int(rng.choice([2, 4, 10]))
        kws = [dict(solver="GAMG", smoother="GaussSeidel", tolerance=1e-7, relTol=0, nCellsInCoarsestLevel=ncl, maxIter=40,
                    nPreSweeps=int(rng.randint(0, 2)), mergeLevels=int(rng.randint(1, 3)))]
        kws.append(dict(solver="PBiCG", preconditioner="DILU", tolerance=1e-6, relTol=0, maxIter=60) if asym else
                   dict(solver="PCG", preconditioner="DIC", tolerance=1e-6, relTol=0, maxIter=60))
        for kw in kws:
            okw = dict(kw)
            if "preconditioner" in okw:
                okw["precond"] = okw.pop("preconditioner")
            xo, po = S.solve(X0, B0, **okw)
            refused = po["nIterations"] < 0     # the oracle refuses what the reference refuses ("No coarse levels created")
            try:
                x, perf = mt.solve(sp["psi"], sp["source"], **kw)
            except Exception as e:
                if not refused:
                    bad.append((n_cases, kw["solver"] + " raised " + str(e)[:120]))
                continue
            if refused:
                bad.append((n_cases, kw["solver"] + " solved what the reference refuses"))
                continue
            n_checks += 1
            h, ho = np.asarray(perf["history"]), np.asarray(po["history"])
            m_ = min(30, h.size, ho.size)
            if perf["nIterations"] != po["nIterations"] and abs(perf["nIterations"] - po["nIterations"]) > max(1, po["nIterations"] // 20):
                bad.append((n_cases, "%s iterations %d vs %d" % (kw["solver"], perf["nIterations"], po["nIterations"])))
            elif not np.allclose(h[:m_], ho[:m_], rtol=1e-5 if kw["solver"] == "PBiCG" else 1e-6, atol=1e-12):
                bad.append((n_cases, kw["solver"] + " history"))
        if ctx.fallback_count():
            bad.append((n_cases, "engine fallback"))
        mt.close(); a.close()
        n_cases += 1
        if bad:
            print("rank %d FAILED case %d (kind %d asym %d nCells %d): %s" % (rank, n_cases - 1, kind, asym, p["nCells"], bad), flush=True)
        allbad = [None] * n
        dist.all_gather_object(allbad, bad)
        if any(allbad):
            bad = bad or [("peer", "another rank failed")]
    ctx.close()
    if rank == 0:
        print(("fuzz FAILED" if bad else "fuzz ok") + ": %d ranks, %d problems, %d comparisons per rank in %.0f s (seed %d)"
              % (n, n_cases, n_checks, time.time() - t0, seed), flush=True)
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
