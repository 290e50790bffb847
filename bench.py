#!/usr/bin/env python
"""bench.py - GAMG / PCG iterations per second and achieved HBM GB/s of the 10 M-cell motorBike p-solve.

Workload (BASELINE.json configs[2]), default `--mesh motorbike_rcm`: simpleFoam's pressure equation on the REAL
motorBike mesh - the reference's own blockMesh + snappyHexMesh on its motorBike.obj, refined to 12 699 795 cells / 38 271 555
internal faces (data/motorbike/mb12.npz, tools/make_motorbike.py), cells renumbered by Foam::bandCompression (renumberMesh) -
solved with the motorBike tutorial's GAMG block (tolerance 1e-7, relTol 0.01, GaussSeidel, nPostSweeps 2, faceAreaPair,
nCellsInCoarsestLevel 10, mergeLevels 1, cacheAgglomeration on).  Where the mesh store is absent the line says so
(config.mesh_fallback) and measures the 216^3 box stand-in of rounds 1-4, which is otherwise a sub-leg (`box216`), like the
same mesh in snappyHexMesh's own numbering, the 8-sub-domain run and the U-equation / PCG legs on the real mesh (`extra`).

A step = one complete p-solve: coefficients (already resident in HBM) handed to the solver
(ldu_matrix_set_coeffs: layout + level-matrix agglomeration, as the reference rebuilds them in
every solver construction), psi reset to 0, then lduMatrix::solver::solve.  One-time addressing
work (dependency levels, agglomeration maps) happens before the timed region, like the
reference's cached lduAddressing / cacheAgglomeration.

value = V-cycles (the reference's nIterations) of all timed steps / wall time; ms per solve and V-cycles per solve stand beside it.
cpu_baseline: the reference's own solver (oracle/_ref) on one host core; cpu_baseline_all_cores: the C restatement with one thread
per sub-domain on all host cores (core count stated).
N > 1: the same matrix decomposed into N sub-domains (strong scaling), one rank per GPU, halo
exchange + scalar all-reduces on RCCL or peer stores.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); what this chip streams is measured: roofline.peak_measured


def kernel_source_hash():
    """hash of the kernel / plan sources a PMC traffic figure belongs to (same function as tools/pmc_traffic.py)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("ldu_kernels.hip", "ldu_cluster.hip", "ldu_internal.hpp", "ldu_plan.cpp"):
        h.update(open(os.path.join(ROOT, "openfoam-2.2.x_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(mesh_spec, kernel, k):
    """HBM bytes per launch of the dominant kernel from a recorded PMC pass (profiles/*_pmc_traffic.json, written by
    tools/pmc_traffic.py: separate rocprofv3 --pmc passes) - quoted only when workload, kernel AND the kernel sources are
    the ones the pass ran on; otherwise (None, why)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if not str(pj.get("workload", "")).startswith(mesh_spec + ",") or pj.get("kernel") != kernel:
            continue
        if "%d pipelined" % k not in pj.get("workload", ""):
            continue
        best = (pj, os.path.basename(f))
    if best is None:
        return None, "no PMC pass recorded for this workload and kernel (tools/pmc_traffic.py)"
    pj, name = best
    if pj.get("source_hash") != kernel_source_hash():
        return None, "profiles/%s is stale: the kernel sources changed since that PMC pass (hash %s, now %s)" % (
            name, pj.get("source_hash"), kernel_source_hash())
    return (pj["bytes_per_launch"], pj.get("bytes_per_launch_corrected")), \
        "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run; traffic = raw counters, " \
        "traffic_corrected = with the gfx950 16-byte correction on the granule polls)" % name

GAMG_CONTROLS = dict(solver="GAMG", tolerance=1e-7, relTol=0.01, smoother="GaussSeidel",
                     nPreSweeps=0, nPostSweeps=2, nFinestSweeps=2, cacheAgglomeration=1,
                     agglomerator="faceAreaPair", nCellsInCoarsestLevel=10, mergeLevels=1)


def resident_pipeline(torch, capi, cases, ctx, addr, mat, n, dev, reps=3):
    """SURVEY 8(f)-1 / VERDICT r2 item 9: simpleFoam's pEqn.H on the box with every field resident in HBM - nothing crosses
    PCIe between assembly, solve and flux (what the fvMatrix glue and the fv kernels exist for):
        rAUf = interpolate(rAU); pEqn = fvm::laplacian(rAUf, p) == fvc::div(phiHbyA), phiHbyA = interpolate(HbyA) & Sf;
        boundary glue (fixedValue outlet, zeroGradient elsewhere); GAMG solve; phi = phiHbyA - pEqn.flux()
    through the C ABI on device pointers only.  Unit-cube hex cells (Sf = unit vectors, weights 1/2, delta 1, V 1).
    Returns V-cycles/s of the whole chain and the share of the time spent outside ldu_solve."""
    import ctypes as C
    L = capi.lib()
    P = capi._ptr
    chk = capi._chk
    l, u, d = cases.box_addressing(n, n, n)
    nC, nF = n ** 3, l.size
    f64 = dict(dtype=torch.float64, device=dev)
    Sf = torch.zeros(nF, 3, **f64)
    Sf[torch.arange(nF, device=dev), torch.from_numpy(d.astype(np.int64)).to(dev)] = 1.0
    negSf = -Sf
    magSf, w, delta, V = (torch.ones(nF, **f64), torch.full((nF,), 0.5, **f64), torch.ones(nF, **f64), torch.ones(nC, **f64))
    c = torch.arange(nC, device=dev, dtype=torch.float64)
    rAU = 1.0 + 0.25 * torch.sin(1e-3 * c)
    HbyA = torch.stack([torch.sin(2e-3 * c), 0.3 * torch.cos(1e-3 * c), 0.1 * torch.sin(3e-3 * c)], dim=1).contiguous()
    # patches: x-min, x-max (outlet: fixedValue 0), y-min, y-max, z-min, z-max
    cc = np.arange(nC, dtype=np.int64)
    i, j, k = cc % n, (cc // n) % n, cc // (n * n)
    fcs = [cc[i == 0], cc[i == n - 1], cc[j == 0], cc[j == n - 1], cc[k == 0], cc[k == n - 1]]
    b = capi.FvBoundary(addr, [x.astype(np.int32) for x in fcs])
    nB = b.n
    iC = torch.zeros(nB, **f64)
    o0, o1 = fcs[0].size, fcs[0].size + fcs[1].size
    # fixedValue outlet: internalCoeffs = -gamma*magSf*deltaCoeffs (fvm::laplacian: gradientInternalCoeffs = -delta,
    # fixedValueFvPatchField.C:108-113, gaussLaplacianScheme.C:75-85), delta = 1/(h/2); boundaryCoeffs = -iC*value = 0
    iC[o0:o1] = -2.0 * rAU[torch.from_numpy(fcs[1]).to(dev)]
    bC = torch.zeros(nB, **f64)                                   # fixedValue 0
    mphiB = torch.zeros(nB, **f64)                                # no flux through the walls / outlet of this test field
    rAUf, gms, diag, upper, mphi, source, psi, fluxI, fluxB, phi = (torch.empty(nF, **f64), torch.empty(nF, **f64),
        torch.empty(nC, **f64), torch.empty(nF, **f64), torch.empty(nF, **f64), torch.empty(nC, **f64), torch.zeros(nC, **f64),
        torch.empty(nF, **f64), torch.empty(nB, **f64), torch.empty(nF, **f64))
    minus1 = torch.full((nF,), -1.0, **f64)
    ctl = capi.make_controls(**GAMG_CONTROLS)
    perf = capi.Perf()

    def assemble():
        chk(L.ldu_fv_interpolate(addr.h, 1, P(w), P(rAU), P(rAUf)))
        chk(L.ldu_fv_faceScale(ctx.h, nF, 1, P(magSf), P(rAUf), 0, P(gms)))
        chk(L.ldu_fvm_laplacian(addr.h, P(delta), P(gms), P(diag), P(upper)))
        chk(L.ldu_fv_interpolateDot(addr.h, 3, P(negSf), P(w), P(HbyA), P(mphi)))          # -phiHbyA
        source.zero_()
        chk(L.ldu_fvm_sourceMinusVDiv(addr.h, b.h, 1, P(mphi), P(mphiB), P(V), P(source)))   # source += V*div(phiHbyA)
        chk(L.ldu_fvm_addBoundaryDiag(b.h, P(iC), P(diag)))
        chk(L.ldu_fvm_addBoundarySource(b.h, P(bC), None, 0, P(source)))

    def solve():
        psi.zero_()
        torch.cuda.synchronize()
        chk(L.ldu_matrix_set_coeffs(mat.h, P(diag), P(upper), None))
        chk(L.ldu_solve(mat.h, C.byref(ctl), P(psi), P(source), C.byref(perf), None))
        return perf.nIterations

    def flux():
        chk(L.ldu_fvm_flux(b.h, P(iC), P(bC), None, P(upper), None, P(psi), P(fluxI), P(fluxB)))
        chk(L.ldu_fv_faceScale(ctx.h, nF, 1, P(minus1), P(mphi), 0, P(phi)))                 # phi = phiHbyA
        chk(L.ldu_fv_faceScale(ctx.h, nF, 1, P(minus1), P(fluxI), 1, P(phi)))                # phi -= flux

    def sync():
        torch.cuda.synchronize(); ctx.sync(); torch.cuda.synchronize()
    assemble(); solve(); flux(); sync()          # warm-up (agglomeration of these face weights is cached on the addressing)
    t_all = t_solve = 0.0
    its = 0
    for _ in range(reps):
        sync(); t0 = time.perf_counter()
        assemble(); sync(); t1 = time.perf_counter()
        its += solve(); sync(); t2 = time.perf_counter()
        flux(); sync(); t3 = time.perf_counter()
        t_all += t3 - t0
        t_solve += t2 - t1
    out = dict(resident_pipeline_vcycles_per_s=round(its / t_all, 2),
               resident_pipeline_assembly_and_flux_share=round(1.0 - t_solve / t_all, 4),
               resident_pipeline_ms=dict(total=round(t_all / reps * 1e3, 3), solve=round(t_solve / reps * 1e3, 3)),
               resident_pipeline_vcycles_per_solve=its // reps,
               resident_pipeline_final_flux_sum=float(phi.sum().item()))
    b.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=216, help="box edge (216 -> 10.08 M cells)")
    ap.add_argument("--mesh", choices=["box", "renumbered", "irregular", "random", "octree", "octree_hexref", "jump2d",
                                       "motorbike", "motorbike_rcm", "motorbike_tiles", "motorbike_real"],
                    default="motorbike_rcm",
                    help="motorbike: the REAL mesh of the metric's workload - the reference's own blockMesh + snappyHexMesh "
                         "(castellatedMesh) on the reference's motorBike.obj, refined to ~10 M cells "
                         "(data/motorbike/<--motorbike-name>.npz, made by tools/make_motorbike.py), in the cell numbering "
                         "snappyHexMesh produced; motorbike_rcm: the same renumbered by Foam::bandCompression (renumberMesh); "
                         "box: the SURVEY 8d C3 stand-in in blockMesh's natural ordering; "
                         "renumbered: the same matrix under Foam::bandCompression (what renumberMesh applies); "
                         "irregular: the box plus random diagonal faces (6-12 neighbours per cell, 3-D locality), "
                         "renumbered by Foam::bandCompression - the unstructured stand-in; "
                         "random: a band-limited random graph (quasi 1-D: ~nC/100 dependency levels, pathological); "
                         "octree: snappyHexMesh-like castellated octree around a motorBike-sized body (hanging faces, "
                         "hexRef8 numbering), renumbered by Foam::bandCompression (openfoam-2.2.x_amd/octree.py); "
                         "octree_hexref: the same in hexRef8's own numbering (what the tutorial's Allrun solves on); "
                         "jump2d: BASELINE config C5's twin at its size - n x n 2-D 5-point matrix with the coefficient "
                         "jumping 1 <-> 1000 across the diagonal (damBreak p_rgh, --n 2000 = 4.0 M cells)")
    ap.add_argument("--real-matrix", default="mb2sl_p3", help="--mesh motorbike_real: which stored matrix (data/motorbike/<name>.npz)")
    ap.add_argument("--tile-size", type=int, default=2048, help="--mesh motorbike_tiles: cells per tile (ldu_tile_shuffle)")
    ap.add_argument("--tile-seed", type=int, default=1, help="--mesh motorbike_tiles: seed of the tile order")
    ap.add_argument("--motorbike-name", default="mb12", help="which stored motorBike mesh (data/motorbike/<name>.npz)")
    ap.add_argument("--octree-q", type=int, default=14, help="octree background mesh 5q x 2q x 2q (14 -> ~10 M cells)")
    ap.add_argument("--octree-levels", type=int, nargs=2, default=[6, 7], help="octree surface refinement levels")
    ap.add_argument("--rank-of", type=int, default=0, metavar="N",
                    help="single-rank PROJECTION of an N-rank run (N = 2, 4, 8) on ONE GPU: rank 0's sub-domain of the N-way "
                         "block decomposition, its processor patches wired to itself over a size-1 RCCL communicator "
                         "(real ncclSend/ncclRecv halo exchanges on the comm stream, real ncclAllReduce for every global "
                         "sum: LDU_FORCE_COMM=1).  Prints the per-rank V-cycle time and the exchanges / all-reduces / "
                         "host read-backs per V-cycle.  Never the headline value: the coupling is to itself, the "
                         "inter-GPU latency is not in it.")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the SAME n^3 matrix decomposed N-way (BASELINE config C4: the 10 M-cell case at "
                         "1/2/4/8 GPUs; default); weak = n^3 cells PER RANK (blocks of a (px n) x (py n) x (pz n) box)")
    ap.add_argument("--comm", choices=["auto", "rccl", "peer"], default="auto",
                    help="N > 1 carrier of halo exchanges and global sums: rccl (ncclSend/ncclRecv + ncclAllReduce), peer "
                         "(values stored into the neighbour's window over xGMI, ldu_peer.hip), auto = measure both, "
                         "cross-check their residual histories, report the faster as `value` and both in `comm_backends`")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than visible GPUs (ranks share GPUs round-robin; peer backend only - RCCL "
                         "refuses two ranks on one device): executes the N-rank path on a 1-GPU box, NOT a scaling number")
    ap.add_argument("--subdomains", type=int, default=0, metavar="K",
                    help="sub-domain mode (ldu_addr_set_subdomains): the matrix cut into K compact sub-domains that are coupled "
                         "like K ranks of the reference (processor-patch semantics: GaussSeidelSmoother.C:98-145, rank-local "
                         "agglomeration) but solved inside ONE context and one set of launches.  A different algorithm from "
                         "the one-rank solve (more V-cycles per solve): its own leg, never the headline")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCG / asymmetric / host-path legs")
    ap.add_argument("--no-sublegs", action="store_true", help="skip the sub-legs of the default line (the 216^3 box stand-in and the "
                                                                "mesh in snappyHexMesh's own numbering, each in its own process)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-n", type=int, default=0, help="box edge of the CPU sample (0 = same)")
    # (ranks spawned by this script receive the command line through the environment: torch.distributed.run's own
    # parser chokes on abbreviable options such as --n behind the script name)
    args = ap.parse_args(json.loads(os.environ["LDU_BENCH_ARGV"]) if "LDU_BENCH_ARGV" in os.environ else None)

    if args.rank_of > 1:
        os.environ["LDU_FORCE_COMM"] = "1"   # read once by the library: every reduction goes through the communicator
        args.no_extras = args.no_cpu = True
    import torch
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand as `python bench.py --gpus N`: become the launcher of N ranks (what the driver does itself with
        # torch.distributed.run); refuse loudly when the box cannot hold them
        import socket
        import subprocess
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not args.oversubscribe:
            raise SystemExit("bench.py: --gpus %d but %d GPU(s) visible (use --oversubscribe to run the %d-rank path on "
                             "shared GPUs: a functional run, not a scaling number)" % (args.gpus, ndev, args.gpus))
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"),
                                                      LDU_BENCH_ARGV=json.dumps(sys.argv[1:]))).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d: start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the lduMatrix HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    shared_gpus = world > ndev
    if shared_gpus and not args.oversubscribe:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (--oversubscribe runs them on shared GPUs)" % (world, ndev))
    if shared_gpus and args.comm == "rccl":
        raise SystemExit("bench.py: RCCL cannot put two ranks on one device; --oversubscribe needs --comm peer")
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    oob_group = None
    if world > 1:
        if shared_gpus:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
            oob_group = dist.new_group(backend="gloo")   # set-up messages of the peer backend (CPU tensors)

    entry.load_package()
    from openfoam_amd import capi, cases, decompose

    n = args.n
    mesh_fallback = None
    if args.mesh.startswith("motorbike"):
        from openfoam_amd import motorbike as _mbchk
        if not _mbchk.available(args.motorbike_name):
            # the metric's own mesh is data outside git (data/motorbike/, 73 MB): say so loudly and measure the stand-in
            mesh_fallback = ("data/motorbike/%s.npz is not on this machine (tools/make_motorbike.py makes it where the reference "
                             "exists): the 216^3 box stand-in was measured instead" % args.motorbike_name)
            sys.stderr.write("bench.py: WARNING: %s\n" % mesh_fallback)
            args.mesh = "box"
    t_gen = time.perf_counter()
    if args.mesh == "random":
        p = cases.random_graph_fast(n ** 3, 7.0, 600)
    elif args.mesh == "jump2d":
        p = cases.jump2d(n, n)
    elif args.mesh == "motorbike_real":
        # the matrix the REFERENCE's simpleFoam handed a p-solve on a snapped + layered motorBike mesh (tools/make_motorbike_matrix.py),
        # cells renumbered by Foam::bandCompression
        from openfoam_amd import motorbike
        p = motorbike.dumped_problem(args.real_matrix)
        mb_meta = p.pop("meta")
        cell_level_hist = mb_meta["internal_faces_per_cell"]
        order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
        nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
        p = cases.renumbered(p, order, fmap, flip, nl, nu)
    elif args.mesh in ("motorbike", "motorbike_rcm", "motorbike_tiles"):
        from openfoam_amd import motorbike
        p = motorbike.problem(args.motorbike_name)
        cell_level_hist = np.bincount(p.pop("cellLevel")).tolist()
        mb_meta = p.pop("meta")
        if args.mesh in ("motorbike_rcm", "motorbike_tiles"):
            order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
            if args.mesh == "motorbike_tiles":
                # bandCompression's order cut into tiles whose order is shuffled (ldu_tile_shuffle): a manualRenumber numbering
                order = capi.tile_shuffle(order, args.tile_size, args.tile_seed)
            nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
            p = cases.renumbered(p, order, fmap, flip, nl, nu)
    elif args.mesh in ("octree", "octree_hexref"):
        from openfoam_amd import octree
        q = args.octree_q
        p = octree.problem(base=(5 * q, 2 * q, 2 * q), surface_levels=tuple(args.octree_levels))
        cell_level_hist = np.bincount(p.pop("cellLevel")).tolist()
        if args.mesh == "octree":
            order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
            nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
            p = cases.renumbered(p, order, fmap, flip, nl, nu)
    elif world > 1 and args.scaling == "weak":
        p = None   # every rank generates its own block (cases.box3d_block)
    else:
        p = cases.irregular_box(n) if args.mesh == "irregular" else cases.box3d(n)
        if args.mesh in ("renumbered", "irregular"):
            order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
            nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
            p = cases.renumbered(p, order, fmap, flip, nl, nu)
    is_octree = args.mesh.startswith("octree") or args.mesh.startswith("motorbike")
    is_mb = args.mesh.startswith("motorbike")
    if world > 1 and args.mesh not in ("box", "motorbike", "motorbike_rcm", "motorbike_tiles"):
        raise SystemExit("bench.py: --mesh %s is a single-GPU measurement" % args.mesh)
    if world > 1 and is_mb and args.scaling == "weak":
        raise SystemExit("bench.py: the motorBike mesh is one mesh: strong scaling only")
    shape = None
    if world > 1 or args.rank_of > 1:
        # 2x2x2 blocks at 8 ranks, slabs/blocks otherwise
        nr = world if world > 1 else args.rank_of
        if args.mesh != "box" and not is_mb:
            raise SystemExit("bench.py: the block decomposition is defined on the box")
        shape = {2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}.get(nr, (1, 1, nr))
    if p is None:
        lp = cases.box3d_block(n, shape, rank)
        nC_total = world * n ** 3
        nF_total = sum(((shape[d] * n - 1) * (shape[(d + 1) % 3] * n) * (shape[(d + 2) % 3] * n)) for d in range(3))
    else:
        nC_total, nF_total = p["nCells"], int(p["lowerAddr"].size)
    t_gen = time.perf_counter() - t_gen
    if p is None:
        pass
    elif is_mb and (world > 1 or args.rank_of > 1):
        # the real mesh, N-way: the reference's own `hierarchical` decomposition of the cell centres where the store holds one for N
        # (tools/make_motorbike.py through the reference's libdecompositionMethods; under a renumbering the stored rank of a cell
        # follows the cell), else equal contiguous ranges of the cell numbering (decomposePar's `simple`-like cut along the
        # numbering; under Foam::bandCompression the ranges are breadth-first shells around the bike).  decomposePar keeps the
        # cell order of the mesh inside a sub-domain, and so does decompose.decompose.
        nr = world if world > 1 else args.rank_of
        shape = (1, 1, nr)
        cell_rank = motorbike.decomposition(args.motorbike_name, nr)
        mb_decomp = "the reference's hierarchical decomposition of the cell centres (motorBike/system/decomposeParDict)"
        if cell_rank is not None and args.mesh != "motorbike":
            cell_rank = cell_rank[order]          # (order[i] = the stored mesh's label of the cell that is cell i now)
        if cell_rank is None:
            cell_rank = (np.arange(p["nCells"], dtype=np.int64) * nr) // p["nCells"]
            mb_decomp = "%d contiguous ranges of the cell numbering" % nr
        # --rank-of N: the LARGEST sub-domain (what an N-rank run waits for)
        my = rank if world > 1 else int(np.argmax(np.bincount(cell_rank, minlength=nr)))
        subs, cell_maps = decompose.decompose(p, cell_rank, nr, only_rank=my)
        lp = subs[my]
        del subs
        if args.rank_of > 1:
            for q in lp["patches_dev"]:
                q["nbrRank"] = 0
    elif world > 1 or args.rank_of > 1:
        cell_rank = decompose.block_ranks(n, n, n, *shape)
        subs, cell_maps = decompose.decompose(p, cell_rank, nr, only_rank=rank)
        lp = subs[rank]
        del subs
        if args.rank_of > 1:
            # every processor patch exchanges with this rank itself (the k-th patch towards rank 0 pairs with the
            # k-th patch of rank 0 towards it: each patch receives its own send buffer through ncclSend/ncclRecv)
            for q in lp["patches_dev"]:
                q["nbrRank"] = 0
    else:
        lp = p
    sub_info = None
    if args.subdomains > 1:
        if world > 1 or args.rank_of > 1:
            raise SystemExit("bench.py: --subdomains is a single-GPU mode")
        t_sd = time.perf_counter()
        sd_rank = decompose.blob_ranks(p["nCells"], p["lowerAddr"], p["upperAddr"], args.subdomains)
        nK = int(sd_rank.max()) + 1       # (a seed that finds only enclosed pockets makes no sub-domain: K can come back smaller)
        lp, sd_order = decompose.concatenated(p, sd_rank, nK)
        sizes = np.bincount(sd_rank, minlength=nK)
        sub_info = dict(K=nK, K_requested=args.subdomains, cells_min=int(sizes.min()), cells_max=int(sizes.max()),
                        interface_faces=int(sum(q["faceCells"].size for q in lp["patches"]) // 2), patches=len(lp["patches"]),
                        decomposition="compact breadth-first blobs of the cell numbering (ldu_partition_blobs)",
                        decomposition_s=round(time.perf_counter() - t_sd, 2))
        args.no_extras = True

    ctx = capi.Context(device_index)
    # carriers: RCCL (one GPU per rank only) and / or the peer-store backend; with both on the context RCCL carries the
    # set-up messages and whatever ctx.comm_select leaves to it.  In `auto` a carrier that cannot be set up on EVERY rank (an
    # image whose RCCL cannot bootstrap, a node whose GPUs cannot map each other's memory) is dropped by all ranks together
    # and said so in the line; the other one carries the run.
    backends = []
    carrier_notes = {}
    if world > 1:
        def everywhere(ok):
            t = torch.tensor([1 if ok else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=oob_group)
            return bool(t.item())
        want_rccl = not shared_gpus and args.comm in ("auto", "rccl")
        want_peer = args.comm in ("auto", "peer")
        for attempt in range(3):
            ok_r = ok_p = True
            try:
                if want_rccl:
                    uid = [capi.Context.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(uid, src=0)
                    ctx.comm_init(rank, world, uid[0])
            except Exception as e:  # pragma: no cover
                ok_r = False
                carrier_notes["rccl"] = str(e)[:200]
            if want_rccl and not everywhere(ok_r):
                if args.comm == "rccl":
                    raise SystemExit("bench.py: the RCCL communicator could not be created: %s" % carrier_notes.get("rccl"))
                carrier_notes.setdefault("rccl", "not available on every rank")
                want_rccl = False
                ctx.close(); ctx = capi.Context(device_index)
                continue
            try:
                if want_peer:
                    ctx.comm_init_peer(rank, world, capi.oob_torch(oob_group))
            except Exception as e:  # pragma: no cover
                ok_p = False
                carrier_notes["peer"] = str(e)[:200]
            if want_peer and not everywhere(ok_p):
                if args.comm == "peer" or not want_rccl:
                    raise SystemExit("bench.py: the peer windows could not be mapped: %s" % carrier_notes.get("peer"))
                carrier_notes.setdefault("peer", "not available on every rank")
                want_peer = False
                ctx.close(); ctx = capi.Context(device_index)
                continue
            break
        backends = (["rccl"] if want_rccl else []) + (["peer"] if want_peer else [])
    elif args.rank_of > 1:
        if args.comm in ("auto", "rccl"):
            ctx.comm_init(0, 1, capi.Context.unique_id())
            backends.append("rccl")
        if args.comm in ("auto", "peer"):
            ctx.comm_init_peer(0, 1)
            backends.append("peer")

    t0 = time.perf_counter()
    addr = capi.Addressing(ctx, lp["nCells"], lp["lowerAddr"], lp["upperAddr"], lp.get("faceWeights"),
                           patches=lp.get("patches_dev", ()), subdomains=lp.get("subdomains"))
    mat = capi.Matrix(addr)
    t_addr = time.perf_counter() - t0
    info = addr.info()

    # inputs resident in HBM before the timed region
    d_diag = torch.from_numpy(lp["diag"]).to(dev)
    d_upper = torch.from_numpy(lp["upper"]).to(dev)
    d_source = torch.from_numpy(lp["source"]).to(dev)
    d_psi = torch.zeros(lp["nCells"], dtype=torch.float64, device=dev)
    d_patch = []
    for i, q in enumerate(lp.get("patches", [])):
        d_patch.append((torch.from_numpy(q["bouCoeffs"]).to(dev), torch.from_numpy(q["intCoeffs"]).to(dev)))
    torch.cuda.synchronize()

    def step(controls):
        d_psi.zero_()
        torch.cuda.synchronize()
        mat.set_coeffs(d_diag, d_upper)
        for i, (b, c) in enumerate(d_patch):
            mat.set_patch_coeffs(i, b, c)
        _, perf = mat.solve(d_psi, d_source, history=True, **controls)
        return perf

    def barrier():
        torch.cuda.synchronize()
        ctx.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # first solve also builds the cached agglomeration (one-time, reported separately)
    t0 = time.perf_counter()
    perf0 = step(GAMG_CONTROLS)
    barrier()
    t_first = time.perf_counter() - t0
    # (the sweep plans of the large levels are built behind the first solves - the level engines sweep meanwhile, same results;
    #  the timed region starts with every plan in place)
    mat.wait_plans()
    t_plans = time.perf_counter() - t0
    for _ in range(max(0, args.warmup - 1)):
        step(GAMG_CONTROLS)

    if os.environ.get("LDU_TRACE_MARKER"):
        # for rocprofv3 summaries of the TIMED region only: a second context runs the placement-census kernel, the
        # marker tools/trace_steady.py cuts the kernel trace at (set-up and warm-up, with their one-time table
        # uploads, lie before it)
        marker_ctx = capi.Context(device_index)

    def timed_region():
        barrier()
        c0 = ctx.comm_counters()
        mat.profile_begin()
        t0 = time.perf_counter()
        its = 0
        pf = None
        for _ in range(args.steps):
            pf = step(GAMG_CONTROLS)
            its += pf["nIterations"]
        barrier()
        el = time.perf_counter() - t0
        pr = mat.profile_end()
        c1 = ctx.comm_counters()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=oob_group)
            el = float(t.item())
        return dict(elapsed=el, iters=its, perf=pf, prof=pr,
                    per_vcycle={k: round((c1[k] - c0[k]) / float(max(1, its)), 2) for k in c1})

    # N > 1 (and the one-rank projection): every carrier on the context is measured; their residual histories must agree
    # (the sums differ in the order of the ranks' partial sums only), `value` is the faster one
    comm_backends = {}
    best = None
    if len(backends) == 2:
        order = ["rccl", "peer"]
    else:
        order = backends or [None]
    for bk in order:
        if bk is not None and len(backends) == 2:
            ctx.comm_select(bk == "peer", bk == "peer")
            step(GAMG_CONTROLS)   # the carrier's own warm-up
        try:
            r = timed_region()
        except Exception as e:  # pragma: no cover
            if bk is None or len(backends) < 2:
                raise
            comm_backends[bk] = dict(error=str(e)[:300])
            continue
        r["backend"] = bk
        if bk is not None:
            comm_backends[bk] = dict(vcycles_per_s=round(r["iters"] / r["elapsed"], 3),
                                     ms_per_vcycle=round(r["elapsed"] / max(1, r["iters"]) * 1e3, 3),
                                     per_vcycle=r["per_vcycle"], carriers=ctx.comm_info(),
                                     residual_history=[float("%.6e" % h) for h in r["perf"]["history"]])
        if best is None or r["iters"] / r["elapsed"] > best["iters"] / best["elapsed"]:
            best = r
    if len(comm_backends) == 2 and all("error" not in v for v in comm_backends.values()):
        ha, hb = (np.array(comm_backends[k]["residual_history"]) for k in ("rccl", "peer"))
        comm_backends["histories_agree"] = bool(ha.size == hb.size and np.allclose(ha, hb, rtol=1e-5, atol=1e-12))
        if not comm_backends["histories_agree"]:
            ctx.comm_select(False, False)   # never report a carrier that disagrees with RCCL
            best = timed_region()
            best["backend"] = "rccl"
    elapsed, iters, perf, prof, comm_per_vcycle = best["elapsed"], best["iters"], best["perf"], best["prof"], best["per_vcycle"]
    comm_carrier = best["backend"]
    # what this chip streams (SURVEY.md 8d "bound"): McCalpin copy / triad with the library's own f64 stream kernels,
    # 10 M and 80 M doubles per array (the 256 MiB Infinity Cache holds the small case: the large one is the HBM figure)
    stream = None
    if rank == 0 and not os.environ.get("LDU_TRACE_MARKER"):   # (not under a kernel trace of the timed region)
        try:
            stream = {"%s_%dM" % (nm, sz // 1000000): round(ctx.stream(md, sz, 20), 1)
                      for sz in (10000000, 80000000) for md, nm in ((0, "copy"), (1, "triad"))}
            stream["unit"] = "GB/s"
        except Exception as e:  # pragma: no cover
            stream = dict(error=str(e)[:200])
    nC, nF = lp["nCells"], int(lp["lowerAddr"].size)
    # algorithmic bytes: GaussSeidel sweep sym 60 nC + 12 nF ; Amul sym 24 nC + 16 nF
    gs_bytes = 60.0 * nC + 12.0 * nF
    amul_bytes = 24.0 * nC + 16.0 * nF
    roof = None
    key, per_launch = ("gs_multi", GAMG_CONTROLS["nFinestSweeps"]) if "gs_multi" in prof else ("gs_sweep", 1)
    if key in prof and prof[key]["count"]:
        ms = prof[key]["ms"] / prof[key]["count"]
        ach = per_launch * gs_bytes / (ms * 1e-3) / 1e9
        eng = addr.sweep_engine(2 if key == "gs_multi" else 1)
        kernels = {"chip-wide point-to-point": ("sweep_p2p_gs_multi_kernel", "sweep_p2p_kernel<SW_GS_FWD>"),
                   "XCD slabs": ("sweep_slab_gs_multi_kernel", "sweep_slab_kernel<SW_GS_FWD>"),
                   "clusters": ("sweep_cluster_gs_multi_kernel", "sweep_cluster_kernel<SW_GS_FWD>"),
                   "single wavefront": ("gs_small_kernel", "gs_small_kernel"),
                   "one workgroup": ("gs_wg_kernel", "gs_wg_kernel"),
                   "blocks": ("gs_blk_kernel", "gs_blk_kernel"),
                   "level kernels": ("sweep_level_kernel", "sweep_level_kernel")}[eng]
        kname = ("%s (%s engine): %d pipelined GaussSeidel sweeps of the finest level per launch"
                 % (kernels[0], eng, per_launch)) if key == "gs_multi" else \
            "%s (%s engine): one GaussSeidel sweep of the rank's finest level" % (kernels[1], eng)
        # traffic: HBM bytes per launch from the PMC counters.  NOT measured by this run: counters need their own
        # rocprofv3 --pmc passes (MI355X_MICROARCH.md); a recorded value is only quoted when workload, kernel and kernel
        # sources are the ones it was recorded for, and the line says where it comes from.
        mesh_spec = {"box": "box:%d" % n, "motorbike": "motorbike:%s" % args.motorbike_name,
                     "motorbike_rcm": "motorbike:%s:rcm" % args.motorbike_name,
                     "motorbike_tiles": "motorbike:%s:tiles%d" % (args.motorbike_name, args.tile_size), "octree": "octree:%d:%d:%d" % (args.octree_q, args.octree_levels[0], args.octree_levels[1]),
                     "octree_hexref": "octree:%d:%d:%d:hexref" % (args.octree_q, args.octree_levels[0], args.octree_levels[1])}.get(args.mesh)
        traffic, traffic_source = (None, None)
        if mesh_spec and world == 1 and key == "gs_multi":
            traffic, traffic_source = pmc_traffic(mesh_spec, kernels[0], per_launch)
        traffic_corrected = None
        if isinstance(traffic, tuple):
            traffic, traffic_corrected = traffic
        roof = dict(bound="hbm", kernel=kname + " (%d dependency levels)" % info["nLevels"],
                    achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                    peak_measured=None, traffic=traffic, traffic_corrected=traffic_corrected, traffic_source=traffic_source, avg_launch_ms=round(ms, 4),
                    bytes_per_launch=per_launch * gs_bytes, launches=prof[key]["count"])
    if roof is not None and stream and "triad_80M" in stream:
        roof["peak_measured"] = stream["triad_80M"]
        roof["frac_of_measured"] = round(roof["achieved"] / stream["triad_80M"], 4)
    amul = None
    if "amul" in prof and prof["amul"]["count"]:
        ms = prof["amul"]["ms"] / prof["amul"]["count"]
        ach = amul_bytes / (ms * 1e-3) / 1e9
        amul = dict(kernel="row_kernel<0> (Amul)", achieved=round(ach, 1), unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 4), avg_launch_ms=round(ms, 4),
                    bytes_per_launch=amul_bytes, launches=prof["amul"]["count"])

    # the reported metric itself against the roofline: one V-cycle by SURVEY.md 8d's formulas with the REAL level
    # sizes (finest 312 nC + 56 nF; every coarse level but the coarsest 232 nC_l + 40 nF_l), over the time of a
    # V-cycle including its share of the per-solve work (coefficient hand-over, level matrices)
    roof_v = None
    levels = []
    try:
        levels = mat.gamg_level_sizes(**GAMG_CONTROLS)
        vbytes = 312.0 * nC + 56.0 * nF + sum(232.0 * L["nCells"] + 40.0 * L["nFaces"] for L in levels[:-1])
        vsec = elapsed / max(1, iters)
        roof_v = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS, bytes_per_vcycle=vbytes,
                      ms_per_vcycle=round(vsec * 1e3, 4), achieved=round(vbytes / vsec / 1e9, 1),
                      frac=round(vbytes / vsec / 1e9 / HBM_PEAK_GBS, 4),
                      levels=[[L["nCells"], L["nFaces"], L["nLevels"], L["engine_gs_multi"]] for L in levels],
                      finest=[nC, nF, info["nLevels"], addr.sweep_engine(2)],
                      cluster_levels_fraction=round(
                          (sum(L["engine_gs_multi"] == "clusters" for L in levels) + (addr.sweep_engine(2) == "clusters"))
                          / float(len(levels) + 1), 3))
    except Exception as e:  # pragma: no cover
        roof_v = dict(error=str(e))

    # extra: PCG+DIC iterations/s on the same matrix (fixed 40 iterations)
    extra = {}
    if args.no_extras:
        extra["skipped"] = True
    else:
      try:
        # what the OpenFOAM shim pays today: diag/upper/source/psi handed over as pageable HOST arrays every solve
        # (PCIe-inclusive; never `value`)
        if world == 1:
            # (only the library calls are timed: the caller's own array handling - zeroing psi - is not the boundary's)
            h_psi = np.zeros(lp["nCells"])
            tH = 0.0
            itsH = 0
            for _ in range(3):
                h_psi[:] = 0.0
                barrier()
                t0 = time.perf_counter()
                mat.set_coeffs(lp["diag"], lp["upper"])
                _, ph = mat.solve(h_psi, lp["source"], history=False, inplace=True, **GAMG_CONTROLS)
                barrier()
                tH += time.perf_counter() - t0
                itsH += ph["nIterations"]
            extra["host_pointer_path_vcycles_per_s"] = round(itsH / tH, 2)
            mat.set_coeffs(d_diag, d_upper)
      except Exception as e:  # pragma: no cover
        extra["host_path_error"] = str(e)
    try:
      if not args.no_extras:
        d_psi.zero_()
        torch.cuda.synchronize()
        mat.profile_begin()
        t0 = time.perf_counter()
        _, pp = mat.solve(d_psi, d_source, history=False, solver="PCG", preconditioner="DIC",
                          tolerance=0.0, relTol=0.0, maxIter=39)
        barrier()
        tp = time.perf_counter() - t0
        pprof = mat.profile_end()
        nit = pp["nIterations"]
        extra["pcg_dic_iterations_per_s"] = round(nit / tp, 2)
        extra["pcg_dic_GBs_algorithmic"] = round((160.0 * nC + 48.0 * nF) * nit / tp / 1e9, 1)
        if "tri_sweep" in pprof:
            extra["dic_sweep_avg_ms"] = round(pprof["tri_sweep"]["ms"] / pprof["tri_sweep"]["count"], 4)
        # DIC preconditioner calls back to back (two half sweeps + the permutations of a C-ABI call each, no host read-back
        # between them as inside the PCG loop; 1.28 ms per call on a box with 0.46 ms half sweeps): on the boxes where the
        # PCG leg runs at ~540 instead of ~735 it/s (profiles/r04_pcg_variance.md) this tells the kernel from the loop
        d_w = torch.zeros_like(d_source)
        Lb = capi.lib()
        capi._chk(Lb.ldu_precondition(mat.h, capi.PRECONDITIONERS["DIC"], capi._ptr(d_w), capi._ptr(d_source), 0))
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            capi._chk(Lb.ldu_precondition(mat.h, capi.PRECONDITIONERS["DIC"], capi._ptr(d_w), capi._ptr(d_source), 0))
        barrier()
        extra["dic_precondition_call_back_to_back_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    except Exception as e:  # pragma: no cover
        extra["pcg_error"] = str(e)
    # config C3's other half: the U-equation solvers of the motorBike case on the ASYMMETRIC matrix of the same box
    # (lower = upper - phi): PBiCG/DILU and the tutorial's own smoothSolver/GaussSeidel
    # (motorBike/system/fvSolution:33-40); fixed iteration counts, it/s and algorithmic GB/s
    if world == 1 and not args.no_extras and args.rank_of <= 1 and args.subdomains <= 1:
        try:
            # (the box: box3d's own asymmetric twin; any other mesh: the same recipe on THIS addressing - cases.asymmetric -
            #  i.e. on the real motorBike mesh for the default line)
            pa = cases.box3d(n, asym=True) if args.mesh == "box" else cases.asymmetric(lp)
            mat.set_coeffs(torch.from_numpy(pa["diag"]).to(dev), torch.from_numpy(pa["upper"]).to(dev),
                           torch.from_numpy(pa["lower"]).to(dev))
            d_srcA = torch.from_numpy(pa["source"]).to(dev)
            for nm, kw, bytes_it in (
                    ("pbicg_dilu", dict(solver="PBiCG", preconditioner="DILU", tolerance=0.0, relTol=0.0, maxIter=19),
                     256.0 * nC + 120.0 * nF),
                    ("smoothsolver_gs", dict(solver="smoothSolver", smoother="GaussSeidel", nSweeps=1, tolerance=0.0,
                                            relTol=0.0, maxIter=20), (60.0 + 24.0 + 8.0) * nC + (20.0 + 24.0) * nF)):
                mat.solve(d_psi.zero_(), d_srcA, history=False, **dict(kw, maxIter=2))     # factors / plans
                d_psi.zero_()
                barrier()
                t0 = time.perf_counter()
                _, pq = mat.solve(d_psi, d_srcA, history=False, **kw)
                barrier()
                tq = time.perf_counter() - t0
                extra[nm + "_iterations_per_s"] = round(pq["nIterations"] / tq, 2)
                extra[nm + "_GBs_algorithmic"] = round(bytes_it * pq["nIterations"] / tq / 1e9, 1)
            del pa, d_srcA
            mat.set_coeffs(d_diag, d_upper)
        except Exception as e:  # pragma: no cover
            extra["asym_error"] = str(e)
        try:
            # north_star's "PCG+GAMG iterations/sec": PCG preconditioned by one GAMG V-cycle (GAMGPreconditioner.C:44-128),
            # 10 iterations of the symmetric p-matrix; bytes per iteration = one V-cycle (roofline_vcycle's formula) + the
            # PCG loop around it (PCG.C:123-172: Amul 24 nC + 16 nF, two dot products 32 nC, p / psi / rA updates 72 nC)
            kwp = dict(solver="PCG", preconditioner="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                       nCellsInCoarsestLevel=10, mergeLevels=1, nVcycles=1, cacheAgglomeration=1, tolerance=0.0, relTol=0.0)
            mat.solve(d_psi.zero_(), d_source, history=False, **dict(kwp, maxIter=1))
            d_psi.zero_()
            barrier()
            t0 = time.perf_counter()
            _, pq = mat.solve(d_psi, d_source, history=True, **dict(kwp, maxIter=9))
            barrier()
            tq = time.perf_counter() - t0
            extra["pcg_gamg_iterations_per_s"] = round(pq["nIterations"] / tq, 2)
            if roof_v and "bytes_per_vcycle" in roof_v:
                extra["pcg_gamg_GBs_algorithmic"] = round((roof_v["bytes_per_vcycle"] + 128.0 * nC + 16.0 * nF) * pq["nIterations"] / tq / 1e9, 1)
            extra["pcg_gamg_residual_after_10"] = float("%.4e" % pq["finalResidual"])
        except Exception as e:  # pragma: no cover
            extra["pcg_gamg_error"] = str(e)

    if world == 1 and args.mesh == "box" and not args.no_extras and args.rank_of <= 1:
        try:
            extra.update(resident_pipeline(torch, capi, cases, ctx, addr, mat, n, dev))
            mat.set_coeffs(d_diag, d_upper)
        except Exception as e:  # pragma: no cover
            extra["resident_pipeline_error"] = str(e)[:300]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:   # the CPU leg is timed at N=1 only
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py
        cn = args.cpu_n or n
        cp = p if (cn == n or is_octree or args.mesh == "jump2d") else cases.box3d(cn)
        scale = 1.0 if (is_octree or args.mesh == "jump2d") else (cn ** 3) / float(n ** 3)
        note = "" if (cn == n or is_octree) else "; scaled by cell count to %d^3" % n
        if oracle_py.ref_available():
            # the reference's own libOpenFOAM (oracle/_ref, built from /root/reference by oracle/build_ref.sh)
            # on this host: the same GAMG p-solve, second of two solves (agglomeration cached like the
            # GPU run's cacheAgglomeration), 1 thread (the reference has no threading, no MPI here)
            d = oracle_py.dict_string(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair",
                                      nCellsInCoarsestLevel=10, mergeLevels=1, cacheAgglomeration="on",
                                      tolerance=1e-7, relTol=0.01)
            res, _ = oracle_py.run_ref("time", dict(cp, psi=np.zeros(cp["nCells"])), d)
            # (own names: `t_first` is the GPU's first solve, reported as extra.first_solve_s - until round 4 this line
            #  overwrote it, and full bench lines carried the REFERENCE's first-solve time there)
            t_ref_first, t_second, it_first, it_second = [float(v) for v in res["time"]]
            cpu = dict(value=round(it_second / t_second * scale, 4), unit="V-cycles/s", cores=1, kind="reference",
                       sample="oracle/_ref/ref_driver = the reference's own lduMatrix::solver (GAMG, GaussSeidel, "
                              "faceAreaPair weights supplied) on %s: second solve %d V-cycles in %.2f s "
                              "(first solve incl. agglomeration %.2f s)%s"
                              % ("the same %s matrix (%d cells)" % (args.mesh, cp["nCells"]) if (is_octree or args.mesh == "jump2d")
                                 else "the %d^3 box" % cn, int(it_second), t_second, t_ref_first, note))
        else:
            S = oracle_py.System(cp)
            okw = dict(smoother="GaussSeidel", nCellsInCoarsestLevel=10, mergeLevels=1,
                       agglomerator="faceAreaPair", tolerance=1e-7, relTol=0.01)
            secs, setup = S.time_gamg_vcycles(cp["source"], nVcycles=2, **okw)
            cpu = dict(value=round(2.0 / secs * scale, 4), unit="V-cycles/s", cores=1, kind="port",
                       sample="oracle (C restatement, gcc -O2, 1 thread): 2 GAMG V-cycles on the %d^3 box "
                              "(%.1f s; agglomeration %.1f s excluded, cached like cacheAgglomeration)"
                              % (cn, secs, setup) + note)

    # the same solve on ALL host cores: the C restatement, box decomposed into one sub-domain per
    # core, one thread per sub-domain emulating the reference's MPI ranks (oracle/time_allcores.py); a port, never
    # the reference itself (no MPI in this image)
    cpu_all = None
    if cpu is not None and not args.no_extras and is_mb:
        # the metric's own mesh on all host cores: as many compact blobs of the numbering as cores, one thread each
        import subprocess
        cores = max(1, min(os.cpu_count() or 1, 64))
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_allcores.py"),
                                "motorbike:%s%s" % (args.motorbike_name, "" if args.mesh != "motorbike" else ":snappy"),
                                str(cores), "2"], capture_output=True, text=True, timeout=900)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            cpu_all = dict(value=round(j["vcycles_per_s"], 4), unit="V-cycles/s", cores=cores, kind="port",
                           sample="oracle (C restatement, gcc -O2 -fopenmp): the same mesh and numbering cut into %d compact blobs "
                                  "(ldu_partition_blobs), one thread per sub-domain emulating the reference's ranks (rank-local "
                                  "GaussSeidel / agglomeration, processor patches, rank-ordered sums): 2 GAMG V-cycles in %.2f s "
                                  "(agglomeration %.1f s excluded, cached like cacheAgglomeration)"
                                  % (j["subdomains"], j["seconds"], j["setup_s"]))
        except Exception as e:  # pragma: no cover
            cpu_all = dict(error=str(e)[:300], cores=cores)
    elif cpu is not None and not args.no_extras and not is_octree and args.mesh != "jump2d":
        import subprocess
        cores = max(1, min(os.cpu_count() or 1, 64))
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_allcores.py"), str(args.cpu_n or n),
                                str(cores), "2"], capture_output=True, text=True, timeout=420)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            sc = ((args.cpu_n or n) ** 3) / float(n ** 3)
            cpu_all = dict(value=round(j["vcycles_per_s"] * sc, 4), unit="V-cycles/s", cores=cores, kind="port",
                           sample="oracle (C restatement, gcc -O2 -fopenmp): %d^3 box decomposed %dx%dx%d, one thread per "
                                  "sub-domain (rank-local GaussSeidel / agglomeration, processor patches, rank-ordered sums): "
                                  "2 GAMG V-cycles in %.2f s (agglomeration %.1f s excluded, cached like cacheAgglomeration)"
                                  % (j["n"], j["blocks"][0], j["blocks"][1], j["blocks"][2], j["seconds"], j["setup_s"]))
        except Exception as e:  # pragma: no cover
            cpu_all = dict(error=str(e)[:300], cores=cores)

    # Sub-legs of the default line, each in its own process (own context, own memory): the 216^3 box stand-in of SURVEY 8d
    # (rounds 1-4's headline; with its PCG / asymmetric / resident-pipeline / host-pointer legs) and the same real mesh in the
    # cell numbering snappyHexMesh wrote (the headline is renumbered by Foam::bandCompression = renumberMesh).
    box_leg = None
    snappy_leg = None
    subdomain_leg = None
    tiles_leg = None
    real_leg = None
    octree_leg = None
    fallbacks_main = ctx.fallback_count()
    mem_in_use_gb = round((lambda fr, tot: (tot - fr) / 1e9)(*torch.cuda.mem_get_info()), 2)
    if rank == 0 and world == 1 and args.mesh == "motorbike_rcm" and not args.no_extras and not args.no_sublegs and args.rank_of <= 1:
        import subprocess
        mat.close(); addr.close(); ctx.close()
        mat = addr = ctx = None
        torch.cuda.empty_cache()

        def run_leg(mesh, extra=()):
            extra = list(extra)
            cpu_flag = [] if "--with-cpu" in extra else ["--no-cpu"]
            extra = [e for e in extra if e != "--with-cpu"]
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--mesh", mesh, "--no-sublegs"] + cpu_flag + extra,
                               capture_output=True, text=True, timeout=900)
            return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        try:
            oj = run_leg("motorbike", ["--motorbike-name", args.motorbike_name, "--no-extras", "--steps", "3", "--warmup", "1"])
            snappy_leg = dict(vcycles_per_s=oj["value"], ms_per_step=oj["ms_per_step"], workload=oj["config"]["workload"],
                              vcycles_per_solve=oj["config"]["vcycles_per_solve"],
                              dependency_levels_finest=oj["config"]["dependency_levels_finest"],
                              finest_launch_ms=oj["roofline"]["avg_launch_ms"], roofline_frac=oj["roofline"]["frac"],
                              roofline_kernel=oj["roofline"]["kernel"], roofline_vcycle_frac=oj["roofline_vcycle"]["frac"],
                              amul_frac=(oj.get("amul") or {}).get("frac"), engine_fallbacks=oj["config"]["engine_fallbacks"],
                              first_solve_s=oj["extra"]["first_solve_s"], residual_history=oj["extra"]["residual_history"])
        except Exception as e:  # pragma: no cover
            snappy_leg = dict(error=str(e)[:300])
        try:
            # the reference's own N-rank semantics inside one GPU: the same matrix cut into 8 sub-domains
            # coupled like 8 ranks.  A different algorithm from the one-rank solve - never compared as V-cycles/s alone:
            # V-cycles per solve and time per solve (to the same tolerance) stand beside it
            oj = run_leg("motorbike_rcm", ["--motorbike-name", args.motorbike_name, "--subdomains", "8", "--steps", "3", "--warmup", "1"])
            subdomain_leg = dict(K=oj["config"]["subdomains"]["K"], subdomains=oj["config"]["subdomains"],
                                 vcycles_per_s=oj["value"], vcycles_per_solve=oj["config"]["vcycles_per_solve"],
                                 ms_per_solve=oj["ms_per_step"], one_domain_ms_per_solve=round(elapsed / args.steps * 1e3, 3),
                                 one_domain_vcycles_per_solve=perf["nIterations"],
                                 dependency_levels=[oj["config"]["dependency_levels_finest"]] + [L[2] for L in oj["roofline_vcycle"]["levels"]],
                                 one_domain_dependency_levels=[info["nLevels"]] + [L["nLevels"] for L in (levels or [])],
                                 engines=[L[3] for L in oj["roofline_vcycle"]["levels"]],
                                 engine_fallbacks=oj["config"]["engine_fallbacks"], residual_history=oj["extra"]["residual_history"],
                                 note="K ranks of the reference inside one context (ldu_addr_set_subdomains): processor-patch "
                                      "coupling between the sub-domains (GaussSeidelSmoother.C:98-145), rank-local agglomeration; "
                                      "patched levels of up to 4 M cells run pipelined on the block engine (interface cells keep two values by sweep "
                                      "parity), larger ones sweep by sweep on the level engines")
        except Exception as e:  # pragma: no cover
            subdomain_leg = dict(error=str(e)[:300])
        try:
            # the numbering is the user's choice (renumberMesh / manualRenumber) and shapes the dependency DAGs of
            # every GAMG level.  The same mesh under bandCompression + shuffled tiles, with the guards: V-cycles per solve, ms per
            # solve to the same tolerance, the residual it stops at, the REFERENCE's CPU time on the same numbering.  The headline
            # stays on bandCompression: the rule "switch when ms per solve drops >= 15 % at an unchanged V-cycle count" is evaluated
            # and printed, and so is why it is not acted on (profiles/r06_numbering.md: the residual after 3 V-cycles sits on the
            # tolerance - 0.985 ... 1.002 x relTol for every tile size and seed tried - so 3 or 4 V-cycles is a coin flip)
            oj = run_leg("motorbike_tiles", ["--motorbike-name", args.motorbike_name, "--no-extras", "--steps", "3", "--warmup", "1",
                                             "--with-cpu"])
            ms_here = elapsed / args.steps * 1e3
            tiles_leg = dict(numbering=oj["config"]["workload"].split(";")[-1].strip(), vcycles_per_s=oj["value"],
                             ms_per_solve=oj["ms_per_step"], vcycles_per_solve=oj["config"]["vcycles_per_solve"],
                             final_residual=oj["extra"]["final_residual"], residual_history=oj["extra"]["residual_history"],
                             bandCompression=dict(ms_per_solve=round(ms_here, 3), vcycles_per_solve=perf["nIterations"],
                                                  final_residual=perf["finalResidual"]),
                             ms_per_decade=round(oj["ms_per_step"] / max(1e-30, -np.log10(oj["extra"]["final_residual"])), 3),
                             bandCompression_ms_per_decade=round(ms_here / max(1e-30, -np.log10(perf["finalResidual"])), 3),
                             dependency_levels=[oj["config"]["dependency_levels_finest"]] + [L[2] for L in oj["roofline_vcycle"]["levels"]],
                             bandCompression_dependency_levels=[info["nLevels"]] + [L["nLevels"] for L in (levels or [])],
                             cpu_reference_same_numbering=oj.get("cpu_baseline"), engine_fallbacks=oj["config"]["engine_fallbacks"],
                             first_solve_s=oj["extra"]["first_solve_s"],
                             rule=dict(text="headline switches only if ms per solve drops >= 15 % at an unchanged V-cycle count",
                                       same_vcycles=bool(oj["config"]["vcycles_per_solve"] == perf["nIterations"]),
                                       ms_ratio=round(oj["ms_per_step"] / ms_here, 4),
                                       acted_on=False,
                                       why_not="the residual after 3 V-cycles is 0.985-1.002 x relTol for every tile size / seed tried "
                                               "(profiles/r06_numbering.md): whether a solve takes 3 or 4 V-cycles is decided by the last "
                                               "per cent; per decade of residual the gain is what ms_per_decade shows"))
        except Exception as e:  # pragma: no cover
            tiles_leg = dict(error=str(e)[:300])
        try:
            # the tutorial's actual kind of mesh and matrix - snapped polyhedra, layer prisms, the coefficients
            # of a real SIMPLE iteration - at 1.85 M cells (the largest the reference's serial snappyHexMesh + simpleFoam make in minutes)
            from openfoam_amd import motorbike as _mbr
            if _mbr.available("mb2sl_p3"):
                oj = run_leg("motorbike_real", ["--no-extras", "--steps", "5", "--warmup", "1"])
                real_leg = dict(workload=oj["config"]["workload"], vcycles_per_s=oj["value"], ms_per_solve=oj["ms_per_step"],
                                vcycles_per_solve=oj["config"]["vcycles_per_solve"], residual_history=oj["extra"]["residual_history"],
                                dependency_levels=[oj["config"]["dependency_levels_finest"]] + [L[2] for L in oj["roofline_vcycle"]["levels"]],
                                engines=[oj["roofline_vcycle"]["finest"][3]] + [L[3] for L in oj["roofline_vcycle"]["levels"]],
                                roofline_vcycle_frac=oj["roofline_vcycle"]["frac"], amul_frac=(oj.get("amul") or {}).get("frac"),
                                engine_fallbacks=oj["config"]["engine_fallbacks"], first_solve_s=oj["extra"]["first_solve_s"])
            else:
                real_leg = dict(skipped="data/motorbike/mb2sl_p3.npz is not on this machine (tools/make_motorbike_matrix.py)")
        except Exception as e:  # pragma: no cover
            real_leg = dict(error=str(e)[:300])
        try:
            oj = run_leg("box", ["--steps", "10", "--warmup", "2"])
            box_leg = dict(vcycles_per_s=oj["value"], ms_per_step=oj["ms_per_step"], workload=oj["config"]["workload"],
                           vcycles_per_solve=oj["config"]["vcycles_per_solve"], roofline=oj["roofline"],
                           roofline_vcycle={k: v for k, v in (oj.get("roofline_vcycle") or {}).items() if k != "levels"},
                           amul=oj.get("amul"), extra=oj.get("extra"), engine_fallbacks=oj["config"]["engine_fallbacks"])
        except Exception as e:  # pragma: no cover
            box_leg = dict(error=str(e)[:300])

    if rank == 0 and args.rank_of > 1:
        # a projection, not a measurement of N GPUs: its own line shape so that nobody mistakes it for the metric
        print(json.dumps({
            "projection": "ONE rank of a %d-rank run on one GPU (bench.py --rank-of %d): %s, %d processor patches (%d faces) "
                          "exchanging with itself (size-1 RCCL communicator / its own "
                          "peer window, LDU_FORCE_COMM=1); inter-GPU latency and load imbalance are NOT in it"
                          % (args.rank_of, args.rank_of,
                             ("the largest sub-domain (%d of %d cells) of the motorBike mesh %s under %s; %s" % (
                                 lp["nCells"], nC_total, args.motorbike_name, args.mesh, mb_decomp)) if is_mb else
                             "rank 0's %d-cell sub-domain of the %d^3 box" % (lp["nCells"], n),
                             len(lp["patches_dev"]), sum(len(q["faceCells"]) for q in lp["patches_dev"])),
            "per_rank_vcycles_per_s": round(iters / elapsed, 3), "ms_per_vcycle": round(elapsed / max(1, iters) * 1e3, 3),
            "vcycles_per_solve": perf["nIterations"], "steps": args.steps,
            "per_vcycle": comm_per_vcycle,
            "carrier": comm_carrier, "comm_backends": comm_backends,
            "engine_fallbacks": ctx.fallback_count(),
            "levels": [[L["nCells"], L["nFaces"], L["nLevels"], L["engine_gs_multi"]] for L in (levels or [])],
            "finest": [nC, nF, info["nLevels"], addr.sweep_engine(1)],
            "residual_history": [float("%.6e" % h) for h in perf["history"]]}))
    elif rank == 0:
        out = {
            "metric": "GAMG p-solve iterations/sec (V-cycles/s) + achieved HBM GB/s, motorBike 10M-cell p-solve" + (
                "" if is_mb else " (stand-in workload, see config.workload)"),
            "value": round(iters / elapsed, 3),
            "unit": "V-cycles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": ("the reference's own simpleFoam matrix and right-hand side (stored)" if args.mesh == "motorbike_real" else
                     "synthetic" + (" coefficients and right-hand side on the REAL mesh (the reference's blockMesh + snappyHexMesh "
                                    "on its motorBike.obj)" if is_mb else "")),
            "config": {"workload": (("simpleFoam motorBike p-solve, the REAL matrix of a SIMPLE iteration: the reference's blockMesh + snappyHexMesh "
                                     "(castellate, SNAP, ADD LAYERS) on motorBike.obj (%s), %d SIMPLE iterations of the reference's "
                                     "simpleFoam, the matrix its last p-solve was handed (laplacian((1|A(U)),p), pEqn.H:14-21); internal faces "
                                     "per cell %s; %d cells, %d faces; GAMG (GaussSeidel, faceAreaPair, tol 1e-7 relTol 0.01), psi = 0"
                                     % (mb_meta["mesh_stages"][-1], mb_meta["iteration"], cell_level_hist, nC_total, nF_total))
                                    if args.mesh == "motorbike_real" else
                                    ("simpleFoam motorBike p-solve on the REAL mesh: the reference's blockMesh + snappyHexMesh "
                                     "(castellatedMesh) on motorBike.obj, background %dx%dx%d, refinementBox level %d, surface "
                                     "levels %d-%d, cells per refinement level %s; %d cells, %d faces; laplacian coefficients "
                                     "|Sf|/(n.d) x (1 + 0.5 u01), fixedValue outlet, GAMG (GaussSeidel, faceAreaPair, tol 1e-7 "
                                     "relTol 0.01)" % (5 * mb_meta["q"], 2 * mb_meta["q"], 2 * mb_meta["q"], mb_meta["box_level"],
                                                       mb_meta["surface_levels"][0], mb_meta["surface_levels"][1],
                                                       cell_level_hist, nC_total, nF_total)) if is_mb else
                                    ("simpleFoam motorBike ~10M-cell p-solve twin: castellated octree (background %dx%dx%d, "
                                     "refinementBox level 4, surface levels %d-%d, hanging faces, cells per refinement level %s; "
                                     "%d cells, %d faces), laplacian coefficients |Sf|/|d| x (1 + 0.5 u01), GAMG "
                                     "(GaussSeidel, faceAreaPair, tol 1e-7 relTol 0.01)"
                                     % (5 * args.octree_q, 2 * args.octree_q, 2 * args.octree_q, args.octree_levels[0],
                                        args.octree_levels[1], cell_level_hist, nC_total, nF_total)) if is_octree else
                                    ("interFoam damBreak p_rgh twin (BASELINE config C5): %d x %d cells 2-D (%d cells, %d faces), "
                                     "GAMG (GaussSeidel, faceAreaPair, tol 1e-7 relTol 0.01)" % (n, n, nC_total, nF_total))
                                    if args.mesh == "jump2d" else
                                    ("simpleFoam motorBike ~10M-cell p-solve stand-in: %s hex box "
                                     "(%d cells, %d faces) variable-coefficient Laplacian, GAMG "
                                     "(GaussSeidel, faceAreaPair, tol 1e-7 relTol 0.01)"
                                     % ("%d^3" % n if p is not None else "%dx%dx%d" % tuple(shape[d] * n for d in range(3)),
                                        nC_total, nF_total)))
                                   + ({"box": "", "renumbered": "; cells renumbered by Foam::bandCompression",
                                       "irregular": "; NOT the plain box: random diagonal faces added (6-12 neighbours per cell), "
                                                    "renumbered by Foam::bandCompression",
                                       "random": "; NOT the box: band-limited random graph, 5-9 neighbours per cell",
                                       "jump2d": "; 5-point matrix, face coefficient jumping 1 <-> 1000 across the diagonal (two-phase density ratio)",
                                       "motorbike": "; cell numbering as snappyHexMesh wrote it",
                                       "motorbike_rcm": "; cells renumbered by Foam::bandCompression (renumberMesh)",
                                       "motorbike_real": "; cells renumbered by Foam::bandCompression (renumberMesh)",
                                       "motorbike_tiles": "; cells renumbered by Foam::bandCompression, then its tiles of %d cells shuffled "
                                                          "(ldu_tile_shuffle, seed %d: a manualRenumber numbering)" % (args.tile_size, args.tile_seed),
                                       "octree": "; hexRef8 cell numbering renumbered by Foam::bandCompression",
                                       "octree_hexref": "; hexRef8 cell numbering (parent keeps its label, 7 children appended)"}[args.mesh]),
                       "mesh": args.mesh,
                       "mesh_fallback": mesh_fallback,
                       "subdomains": sub_info,
                       "parallelism": ("domain decomposition x%d" % world) + (
                           "" if world == 1 else " (%s, %s scaling: %s; halo exchanges and global sums by %s%s)" % (
                               mb_decomp if is_mb else
                               "x".join(str(v) for v in shape) + " blocks", args.scaling,
                               "the same mesh" if is_mb else
                               ("%d^3 cells per rank" % n) if args.scaling == "weak" else "the same %d^3 matrix" % n,
                               {"rccl": "RCCL (ncclSend/ncclRecv, ncclAllReduce)", "peer": "peer stores into the neighbour's "
                                "window over xGMI (ldu_peer.hip)"}.get(comm_carrier, comm_carrier),
                               "; RANKS SHARE GPUS (--oversubscribe): a functional run of the N-rank path, not a scaling number"
                               if shared_gpus else "")),
                       "rccl_ranks": ctx.comm_info()["rccl_ranks"] if world > 1 else 0,
                       "gpus_visible": ndev,
                       "comm_per_vcycle": comm_per_vcycle if world > 1 else None,
                       "comm_backends": comm_backends or None,
                       "carriers_dropped": carrier_notes or None,
                       "vcycles_per_solve": perf["nIterations"],
                       "dependency_levels_finest": info["nLevels"],
                       # sweeps that expired a dependency wait and were re-run on the level-kernel engine (0 = the fast
                       # engines carried every sweep of the timed region)
                       "engine_fallbacks": fallbacks_main},
            "roofline": roof,
            "stream": stream,
            "roofline_vcycle": roof_v,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "amul": amul,
            "box216": box_leg,
            "motorbike_snappy_numbering": snappy_leg,
            "subdomains_8": subdomain_leg,
            "motorbike_tile_numbering": tiles_leg,
            "motorbike_snapped_layered_real_matrix": real_leg,
            "extra": dict(extra, device_memory_in_use_GB=mem_in_use_gb,
                          first_solve_s=round(t_first, 3), plans_ready_s=round(t_plans, 3), addressing_setup_s=round(t_addr, 3),
                          problem_generation_s=round(t_gen, 3),
                          initial_residual=perf["initialResidual"], final_residual=perf["finalResidual"],
                          residual_history=[float("%.6e" % h) for h in perf["history"]]),
        }
        print(json.dumps(out))
    if mat is not None:
        mat.close()
        addr.close()
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
