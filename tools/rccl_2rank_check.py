"""Two (or N) processes, one GPU each, REAL RCCL: run under
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/rccl_2rank_check.py
Each rank takes its sub-domain of a decomposed box (processor patches), initialises the library's RCCL communicator
(ldu_ctx_comm_init, unique id broadcast over gloo) and runs Amul / residual / GaussSeidel / PCG-DIC / GAMG through
csrc/ldu_comm.cpp (halo exchange on the comm stream, device-resident all-reduces).  Rank 0 gathers and compares with
the oracle's serial emulation of the same N-rank algorithm.  Exit code 0 = all equal.  Used by
tests/test_gpu_multidomain.py::test_rccl_two_processes (skipped with fewer than 2 GPUs)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import conftest  # noqa: F401,E402
from openfoam_amd import capi, cases, decompose  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
dist.init_process_group("gloo")
n = 16
p = cases.box3d(n)
shape = {2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}.get(world, (1, 1, world))
cr = decompose.block_ranks(n, n, n, *shape)
subs, maps = decompose.decompose(p, cr, world)
sp = subs[rank]
ctx = capi.Context(local)
uid = [capi.Context.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
ctx.comm_init(rank, world, uid[0])
a = capi.Addressing(ctx, sp["nCells"], sp["lowerAddr"], sp["upperAddr"], sp.get("faceWeights"), patches=sp["patches_dev"])
m = capi.Matrix(a)
m.set_coeffs(sp["diag"], sp["upper"])
for i, q in enumerate(sp["patches"]):
    m.set_patch_coeffs(i, q["bouCoeffs"], q["intCoeffs"])
rng = np.random.RandomState(5)
xs = [rng.randn(s["nCells"]) for s in subs]
bs = [rng.randn(s["nCells"]) for s in subs]
res = dict(Amul=m.Amul(xs[rank]), res=m.residual(xs[rank], bs[rank]), gs=m.smooth("GaussSeidel", xs[rank], bs[rank], 2))
kw = dict(tolerance=1e-9, relTol=0)
res["pcg"], pp = m.solve(sp["psi"], sp["source"], solver="PCG", preconditioner="DIC", **kw)
res["pcg_hist"] = pp["history"]
gk = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10, mergeLevels=1,
          tolerance=1e-8, relTol=0)
res["gamg"], pg = m.solve(sp["psi"], sp["source"], **gk)
res["gamg_hist"] = pg["history"]
res["overlapped"] = ctx.overlapped_halo_count()
allres = [None] * world
dist.gather_object(res, allres if rank == 0 else None, dst=0)
rc = 0
if rank == 0:
    import oracle_py as O
    S = O.System(subs)
    X, B = np.concatenate(xs), np.concatenate(bs)
    cat = lambda k: np.concatenate([r[k] for r in allres])
    checks = dict(Amul=np.array_equal(cat("Amul"), S.Amul(X)), residual=np.array_equal(cat("res"), S.residual(X, B)),
                  GaussSeidel=np.array_equal(cat("gs"), S.smooth("GaussSeidel", X, B, 2)))
    X0, SRC = np.concatenate([s["psi"] for s in subs]), np.concatenate([s["source"] for s in subs])
    xo, po = S.solve(X0, SRC, solver="PCG", precond="DIC", **kw)
    checks["PCG its"] = len(allres[0]["pcg_hist"]) == len(po["history"])
    checks["PCG hist"] = checks["PCG its"] and np.allclose(allres[0]["pcg_hist"], po["history"], rtol=1e-6, atol=1e-12)
    checks["PCG x"] = np.max(np.abs(cat("pcg") - xo)) <= 1e-8 * np.max(np.abs(xo))
    ok = dict(gk); ok.pop("solver")
    xg, pg0 = S.solve(X0, SRC, solver="GAMG", **ok)
    checks["GAMG its"] = len(allres[0]["gamg_hist"]) == len(pg0["history"])
    checks["GAMG hist"] = checks["GAMG its"] and np.allclose(allres[0]["gamg_hist"], pg0["history"], rtol=1e-6, atol=1e-12)
    checks["GAMG x"] = np.max(np.abs(cat("gamg") - xg)) <= 1e-8 * np.max(np.abs(xg))
    checks["halo exchanges overlapped"] = all(r["overlapped"] > 0 for r in allres)
    print("rccl_2rank_check (%d ranks):" % world, checks, flush=True)
    rc = 0 if all(checks.values()) else 1
m.close(); a.close(); ctx.close()
t = torch.tensor([rc])
dist.broadcast(t, src=0)
dist.destroy_process_group()
sys.exit(int(t.item()))
