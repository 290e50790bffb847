"""TEST INFRASTRUCTURE / bench.py cpu_baseline leg - the CPU restatement of the GAMG p-solve on ALL host cores
(SURVEY.md 8d "CPU reference timing": domain-decomposed, threads emulating ranks, core count stated).
The n^3 box of bench.py is decomposed into `cores` sub-domains with processor patches (what decomposePar gives the
reference's MPI ranks); libldu_oracle_omp.so runs every rank-local loop one sub-domain per thread, halo values and
global sums exactly as the serial emulation.  Prints one JSON object.
  python oracle/time_allcores.py <n> <cores> [nVcycles=2]
  python oracle/time_allcores.py motorbike:<stored mesh>[:snappy] <cores> [nVcycles=2]
      the stored motorBike mesh (renumbered by Foam::bandCompression unless :snappy, as bench.py does), cut into `cores` compact
      breadth-first blobs of the cell numbering (ldu_partition_blobs: what a graph decomposition method would hand the ranks)"""
import json
import os
import sys
import time

spec = sys.argv[1]
n, cores = (int(spec) if spec.isdigit() else 0), int(sys.argv[2])
nV = int(sys.argv[3]) if len(sys.argv) > 3 else 2
os.environ["LDU_ORACLE_OMP"] = "1"
os.environ["OMP_NUM_THREADS"] = str(cores)
os.environ.setdefault("OMP_PROC_BIND", "spread")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np  # noqa: E402
import __graft_entry__ as entry  # noqa: E402
entry.load_package()
from openfoam_amd import cases, decompose  # noqa: E402
import oracle_py  # noqa: E402

if not spec.isdigit():
    from openfoam_amd import capi, motorbike  # noqa: E402
    f = spec.split(":")
    p = motorbike.problem(f[1])
    p.pop("cellLevel"); p.pop("meta")
    if not (len(f) > 2 and f[2] == "snappy"):
        order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
        nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
        p = cases.renumbered(p, order, fmap, flip, nl, nu)
    t0 = time.perf_counter()
    rank = decompose.blob_ranks(p["nCells"], p["lowerAddr"], p["upperAddr"], cores)
    K = int(rank.max()) + 1
    subs, _ = decompose.decompose(p, rank, K)
    t_dec = time.perf_counter() - t0
    S = oracle_py.System(subs)
    src = np.concatenate([s["source"] for s in subs])
    okw = dict(smoother="GaussSeidel", nCellsInCoarsestLevel=10, mergeLevels=1, agglomerator="faceAreaPair",
               tolerance=1e-7, relTol=0.01)
    secs, setup = S.time_gamg_vcycles(src, nVcycles=nV, **okw)
    print(json.dumps(dict(vcycles_per_s=nV / secs, seconds=secs, setup_s=setup, decompose_s=t_dec, cores=cores, subdomains=K,
                          cells=int(p["nCells"]), mesh=spec, nVcycles=nV)))
    sys.exit(0)
p = cases.box3d(n)
# near-cubic block decomposition of `cores` ranks: factor cores into px*py*pz, largest factor along z
f = [1, 1, 1]
c = cores
for prime in (2, 3, 5, 7, 11, 13):
    while c % prime == 0:
        f[int(np.argmin(f))] *= prime
        c //= prime
if c > 1:
    f[int(np.argmin(f))] *= c
px, py, pz = sorted(f)
t0 = time.perf_counter()
subs, _ = decompose.decompose(p, decompose.block_ranks(n, n, n, px, py, pz), cores)
t_dec = time.perf_counter() - t0
S = oracle_py.System(subs)
src = np.concatenate([s["source"] for s in subs])
okw = dict(smoother="GaussSeidel", nCellsInCoarsestLevel=10, mergeLevels=1, agglomerator="faceAreaPair",
           tolerance=1e-7, relTol=0.01)
secs, setup = S.time_gamg_vcycles(src, nVcycles=nV, **okw)
print(json.dumps(dict(vcycles_per_s=nV / secs, seconds=secs, setup_s=setup, decompose_s=t_dec, cores=cores,
                      blocks=[px, py, pz], n=n, nVcycles=nV)))
