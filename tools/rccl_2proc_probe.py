"""N-process RCCL smoke test of the multi-rank path (halo ncclSend/Recv, scalar allreduce, GAMG interface
set-up) for a box with >= N GPUs.  On a 1-GPU box RCCL refuses two ranks on one device
("invalid usage": measured, round 1), so this cannot run under gpurun; the same call sites are covered by
the threaded local communicator in tests/test_gpu_multidomain.py.
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_2proc_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import __graft_entry__ as entry

entry.load_package()
from openfoam_amd import capi, cases, decompose

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
ndev = torch.cuda.device_count()
ctx = capi.Context(int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev))
uid = [capi.Context.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
ctx.comm_init(rank, world, uid[0])
p = cases.box3d(24)
cr = decompose.block_ranks(24, 24, 24, 1, 1, world)
subs, maps = decompose.decompose(p, cr, world, only_rank=rank)
sp = subs[rank]
a = capi.Addressing(ctx, sp["nCells"], sp["lowerAddr"], sp["upperAddr"], sp.get("faceWeights"), patches=sp["patches_dev"])
m = capi.Matrix(a)
m.set_coeffs(sp["diag"], sp["upper"])
for i, q in enumerate(sp["patches"]):
    m.set_patch_coeffs(i, q["bouCoeffs"], q["intCoeffs"])
x, perf = m.solve(sp["psi"], sp["source"], solver="GAMG", smoother="GaussSeidel", tolerance=1e-8, relTol=0)
print("rank", rank, "GAMG", perf["nIterations"], perf["finalResidual"], flush=True)
x, perf = m.solve(sp["psi"], sp["source"], solver="PCG", preconditioner="DIC", tolerance=1e-8, relTol=0)
print("rank", rank, "PCG", perf["nIterations"], perf["finalResidual"], flush=True)
dist.barrier()
