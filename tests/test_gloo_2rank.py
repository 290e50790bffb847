"""world_size-2 gloo test on CPU of the N>1 host path: the decomposition bench.py hands to the
ranks (decompose.decompose(only_rank=...)), the processor-patch list the library consumes (`patches_dev`, in
patch order) and the communication schedule libldugpu issues on RCCL for one operator application
(csrc/ldu_solvers.cpp dev_amul + csrc/ldu_comm.cpp comm_exchange): pack all patches into one send buffer ->
start send/recv per patch in patch order (k-th send to a rank pairs with its k-th receive) -> interior rows while
the exchange is in flight -> wait -> apply the received values to the boundary rows; one all-reduce per global sum.
The RCCL calls themselves need GPUs (tests/test_gpu_multidomain.py: test_rccl_halo_exchange_on_one_rank,
test_rccl_two_processes).  Each rank runs a numpy PCG over torch.distributed/gloo, once with the diagonal
preconditioner (decomposition-invariant: must equal the undecomposed solve) and once with the RANK-LOCAL DIC
preconditioner of the reference's parallel runs (decomposition-dependent: must equal the oracle's serial emulation
of the 2-rank algorithm); rank 0 compares."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _amul(sp, x):
    """dev_amul: pack -> exchange started -> interior rows -> wait -> apply (lduMatrixATmul.C:34-92)"""
    dev = sp["patches_dev"]                       # what capi.Addressing(patches=...) receives
    offs = np.concatenate([[0], np.cumsum([len(q["faceCells"]) for q in dev])]).astype(int)
    send_all = np.concatenate([x[q["faceCells"]] for q in dev]) if dev else np.zeros(0)   # pack_kernel
    recv_all = torch.zeros(send_all.size, dtype=torch.float64)
    reqs = []
    # comm_exchange: grouped send/recv in the LIBRARY's issue order (ldu_comm.cpp comm_remote_order through the C ABI:
    # a change of that order, or of which patches are exchanged, changes what this test sends)
    from openfoam_amd import capi
    order = capi.comm_exchange_order([len(q["faceCells"]) for q in dev], [-1] * len(dev))
    assert sorted(order.tolist()) == list(range(len(dev)))
    for i in order.tolist():
        q = dev[i]
        reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(send_all[offs[i]:offs[i + 1]])), q["nbrRank"]))
        reqs.append(dist.irecv(recv_all[offs[i]:offs[i + 1]], q["nbrRank"]))
    l, u = sp["lowerAddr"], sp["upperAddr"]
    y = sp["diag"] * x                            # interior rows, overlapped with the exchange
    np.add.at(y, u, sp["upper"] * x[l])
    np.add.at(y, l, sp["upper"] * x[u])
    for rq in reqs:                               # comm_wait_halo
        rq.wait()
    recv = recv_all.numpy()
    for i, q in enumerate(sp["patches"]):         # apply_patches: result[faceCells] -= bouCoeffs*psiNbr
        np.subtract.at(y, q["faceCells"], q["bouCoeffs"] * recv[offs[i]:offs[i + 1]])
    return y


def _dic_factor(sp):
    """DICPreconditioner::calcReciprocalD (DICPreconditioner.C:57-84), rank-local: the interfaces do not enter"""
    rD = sp["diag"].copy()
    l, u, up = sp["lowerAddr"], sp["upperAddr"], sp["upper"]
    for f in range(l.size):
        rD[u[f]] -= up[f] * up[f] / rD[l[f]]
    return 1.0 / rD


def _dic_apply(sp, rD, r):
    """DICPreconditioner::precondition (DICPreconditioner.C:87-123)"""
    l, u, up = sp["lowerAddr"], sp["upperAddr"], sp["upper"]
    w = rD * r
    for f in range(l.size):
        w[u[f]] -= rD[u[f]] * up[f] * w[l[f]]
    for f in range(l.size - 1, -1, -1):
        w[l[f]] -= rD[l[f]] * up[f] * w[u[f]]
    return w


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
    assert [q["nbrRank"] for q in sp["patches_dev"]] == [1 - rank]       # one processor patch towards the other rank
    # the pairing rule of the library (ldu_comm.cpp paired_patch) against what decomposePar-style pairing means: both
    # ranks list the same global faces on paired patches.  Patch lists travel over gloo like the restrict maps do.
    from openfoam_amd import capi
    mine = [dict(nbr=q["nbrRank"], faces=np.asarray(q["faces"])) for q in sp["patches"]]
    lists = [None] * world
    dist.all_gather_object(lists, mine)
    for pi, q in enumerate(mine):
        theirs = lists[q["nbr"]]
        j = capi.comm_paired_patch([m_["nbr"] for m_ in mine], pi, [t["nbr"] for t in theirs], rank)
        assert j >= 0 and np.array_equal(theirs[j]["faces"], q["faces"]), (rank, pi, j)
    res = {}
    for pre in ("diagonal", "DIC"):
        x = np.zeros(sp["nCells"]); b = sp["source"]
        rD = 1.0 / sp["diag"] if pre == "diagonal" else _dic_factor(sp)
        r = b - _amul(sp, x)
        norm = _gsum(np.sum(np.abs(b)))
        hist = [_gsum(np.sum(np.abs(r))) / norm]
        pvec = None
        rho_old = 1.0
        for it in range(200):
            w = rD * r if pre == "diagonal" else _dic_apply(sp, rD, r)
            rho = _gsum(float(w @ r))
            pvec = w if pvec is None else w + (rho / rho_old) * pvec
            Ap = _amul(sp, pvec)
            alpha = rho / _gsum(float(Ap @ pvec))
            x += alpha * pvec
            r -= alpha * Ap
            rho_old = rho
            hist.append(_gsum(np.sum(np.abs(r))) / norm)
            if hist[-1] < 1e-8:
                break
        res[pre] = (x, hist)
    out[rank] = (res["diagonal"][0], res["diagonal"][1], maps[rank], res["DIC"][0], res["DIC"][1])
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
    # rank-local DIC (the reference's parallel semantics): equals the oracle's serial emulation of the
    # 2-rank algorithm - and differs from the undecomposed DIC-PCG, whose preconditioner sees the cut faces
    xd, pd = oracle.System(subs).solve(X0, B, solver="PCG", precond="DIC", tolerance=1e-8, relTol=0)
    histd = np.array(out[0][4])
    assert len(histd) - 1 == pd["nIterations"]
    np.testing.assert_allclose(histd, pd["history"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(np.concatenate([out[r][3] for r in range(world)]), xd, rtol=1e-7, atol=1e-9)
    x2, p2 = oracle.System(p).solve(p["psi"], p["source"], solver="PCG", precond="DIC", tolerance=1e-8, relTol=0)
    assert not np.allclose(histd[:len(p2["history"])][1:4], p2["history"][1:4], rtol=1e-6)


def test_pairing_rule_with_several_patches_per_rank_pair():
    """ldu_comm.cpp paired_patch: the k-th patch of rank b towards a pairs with the k-th patch of rank a towards b
    (two patches between one rank pair: a ring of four slabs on two ranks, processorCyclic-like)."""
    from openfoam_amd import capi
    a = [1, 2, 1, 3, 1]          # rank 0's patches: neighbour ranks
    b = [0, 0, 4, 0]             # rank 1's patches
    assert [capi.comm_paired_patch(a, p, b, 0) for p in (0, 2, 4)] == [0, 1, 3]
    assert [capi.comm_paired_patch(b, p, a, 1) for p in (0, 1, 3)] == [0, 2, 4]
    assert capi.comm_paired_patch(a, 3, [5, 6], 0) == -1     # rank 3 lists no patch towards rank 0: unpaired
    # exchange order: patches with faces that are not cyclic, ascending
    assert capi.comm_exchange_order([5, 0, 3, 7], [-1, -1, 3, -1]).tolist() == [0, 3]


def _oob_worker(rank, world, port, out):
    """the out-of-band exchange the peer-store backend uses for its set-up messages (capi.oob_torch: isend / irecv of
    byte blobs over gloo) + the weak-scaling block generator: every rank builds its own block and checks through the
    exchange that both sides of every cut face computed the same coefficient"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest  # noqa: F401
    from openfoam_amd import capi, cases
    ex = capi.oob_torch()
    # 1. blobs of different sizes to every other rank (what ldu_ctx_comm_init_peer does with the window handles)
    peers = [r for r in range(world) if r != rank]
    got = ex(peers, [bytes([rank]) * (10 + p) for p in peers], [10 + rank] * len(peers))
    ok = all(g == bytes([p]) * (10 + rank) for p, g in zip(peers, got))
    # 2. the patch coefficients of the blocks, pairwise (what comm_peer_setup_addr / comm_exchange_ints do per neighbour)
    shape = {2: (1, 1, 2), 4: (1, 2, 2)}[world]
    sp = cases.box3d_block(5, shape, rank)
    nb = [q["nbrRank"] for q in sp["patches"]]
    send = [q["bouCoeffs"].tobytes() for q in sp["patches"]]
    recv = ex(nb, send, [len(s) for s in send])
    same = all(np.array_equal(np.frombuffer(r, dtype=np.float64), q["bouCoeffs"]) for r, q in zip(recv, sp["patches"]))
    out[rank] = (ok, same, len(nb))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_out_of_band_exchange_and_block_generator(world):
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29950 + (os.getpid() % 40) + world
    mp.spawn(_oob_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        ok, same, n = out[r]
        assert ok and same and n >= 1, (r, out[r])


def test_weak_scaling_blocks_assemble_to_one_symmetric_operator():
    """cases.box3d_block: the 8 blocks of a 2 x 2 x 2 decomposition assemble (diag + internal faces + patch coefficients) to a
    symmetric matrix whose A x* is the concatenated source - without ever building the global matrix on a rank"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from openfoam_amd import cases
    n, blocks = 5, (2, 2, 2)
    subs = [cases.box3d_block(n, blocks, r) for r in range(8)]
    NX = NY = NZ = 2 * n

    def gid(r):
        bi, bj, bk = r % 2, (r // 2) % 2, r // 4
        c = np.arange(n ** 3)
        i, j, k = c % n, (c // n) % n, c // (n * n)
        return (bi * n + i) + NX * ((bj * n + j) + NY * (bk * n + k))
    nC = NX * NY * NZ
    A = np.zeros((nC, nC)); b = np.zeros(nC)
    for r, s in enumerate(subs):
        g = gid(r)
        A[g, g] += s["diag"]
        A[g[s["lowerAddr"]], g[s["upperAddr"]]] += s["upper"]
        A[g[s["upperAddr"]], g[s["lowerAddr"]]] += s["upper"]
        b[g] = s["source"]
        assert [q["nbrRank"] for q in s["patches"]] == sorted(q["nbrRank"] for q in s["patches"])
        for q in s["patches"]:
            q2 = [t for t in subs[q["nbrRank"]]["patches"] if t["nbrRank"] == r][0]
            assert np.array_equal(q["bouCoeffs"], q2["bouCoeffs"])
            A[g[q["faceCells"]], gid(q["nbrRank"])[q2["faceCells"]]] += -q["bouCoeffs"]
    assert np.allclose(A, A.T)
    assert np.abs(A @ np.sin(1e-3 * np.arange(nC)) - b).max() < 1e-12
