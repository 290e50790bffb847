"""world_size-2 gloo test on CPU of the N>1 host path: the decomposition bench.py hands to the
ranks (decompose.decompose(only_rank=...)), the processor-patch pairing and the communication
schedule of one PCG iteration (pack -> exchange -> apply, one all-reduce per global sum) - the
same schedule libldugpu issues on RCCL.  Each rank runs a small numpy PCG (diagonal
preconditioner, decomposition-invariant) over torch.distributed/gloo; rank 0 compares with the
oracle's serial emulation of the 2-rank run and with the undecomposed solve."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _amul(sp, x, recv):
    l, u = sp["lowerAddr"], sp["upperAddr"]
    y = sp["diag"] * x
    np.add.at(y, u, sp["upper"] * x[l])
    np.add.at(y, l, sp["upper"] * x[u])
    for q, r in zip(sp["patches"], recv):
        np.subtract.at(y, q["faceCells"], q["bouCoeffs"] * r)   # result[faceCells] -= bouCoeffs*psiNbr
    return y


def _exchange(sp, x):
    reqs, recv = [], []
    for q in sp["patches"]:
        send = torch.from_numpy(np.ascontiguousarray(x[q["faceCells"]]))
        r = torch.zeros(len(q["faceCells"]), dtype=torch.float64)
        reqs.append(dist.isend(send, q["nbrRank"]))
        reqs.append(dist.irecv(r, q["nbrRank"]))
        recv.append(r)
    for rq in reqs:
        rq.wait()
    return [r.numpy() for r in recv]


def _gsum(v):
    t = torch.tensor([v], dtype=torch.float64)
    dist.all_reduce(t)
    return float(t.item())


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest  # noqa: F401
    from openfoam_amd import cases, decompose
    p = cases.box3d(8)
    cr = decompose.block_ranks(8, 8, 8, 1, 1, world)
    subs, maps = decompose.decompose(p, cr, world, only_rank=rank)
    sp = subs[rank]
    x = np.zeros(sp["nCells"]); b = sp["source"]
    rD = 1.0 / sp["diag"]
    r = b - _amul(sp, x, _exchange(sp, x))
    norm = _gsum(np.sum(np.abs(b)))
    hist = [_gsum(np.sum(np.abs(r))) / norm]
    pvec = None
    rho_old = 1.0
    for it in range(200):
        w = rD * r
        rho = _gsum(float(w @ r))
        pvec = w if pvec is None else w + (rho / rho_old) * pvec
        Ap = _amul(sp, pvec, _exchange(sp, pvec))
        alpha = rho / _gsum(float(Ap @ pvec))
        x += alpha * pvec
        r -= alpha * Ap
        rho_old = rho
        hist.append(_gsum(np.sum(np.abs(r))) / norm)
        if hist[-1] < 1e-8:
            break
    out[rank] = (x, hist, maps[rank])
    dist.destroy_process_group()


def test_two_rank_gloo_pcg(oracle):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    from openfoam_amd import cases, decompose
    p = cases.box3d(8)
    cr = decompose.block_ranks(8, 8, 8, 1, 1, world)
    subs, maps = decompose.decompose(p, cr, world)
    X0 = np.concatenate([s["psi"] for s in subs]); B = np.concatenate([s["source"] for s in subs])
    xo, po = oracle.System(subs).solve(X0, B, solver="PCG", precond="diagonal", tolerance=1e-8, relTol=0)
    x = np.concatenate([out[r][0] for r in range(world)])
    hist = np.array(out[0][1])
    assert len(hist) - 1 == po["nIterations"]
    # psi0 = 0 -> normFactor = sum|b| exactly as in lduMatrix::solver::normFactor
    np.testing.assert_allclose(hist, po["history"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(x, xo, rtol=1e-7, atol=1e-9)
    # and the decomposed run equals the undecomposed one (diagonal PCG is decomposition-invariant)
    x1, p1 = oracle.System(p).solve(p["psi"], p["source"], solver="PCG", precond="diagonal",
                                    tolerance=1e-8, relTol=0)
    full = np.zeros(p["nCells"])
    for r in range(world):
        full[out[r][2]] = out[r][0]
    np.testing.assert_allclose(full, x1, rtol=1e-6, atol=1e-8)
