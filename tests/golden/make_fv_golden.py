"""Generates tests/golden/fv_*.npz: inputs + outputs of the REFERENCE's own finite-volume stencils
(oracle/_ref/fv_driver = libfiniteVolume units compiled from /root/reference by oracle/build_ref_fv.sh)
on perturbed, graded hex boxes.  Run here (the reference does not travel); the .npz files are data:
addressing, geometry, input fields, reference outputs.   python tests/golden/make_fv_golden.py"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fv_case  # noqa: E402

CASES = {"fv_box_7x6x5": (7, 6, 5, 3), "fv_box_12x3x9": (12, 3, 9, 8)}


def generate(name):
    nx, ny, nz, seed = CASES[name]
    mesh = fv_case.box_mesh(nx, ny, nz, seed=seed)
    rng = np.random.RandomState(100 + seed)
    nC, nF = mesh["nCells"], mesh["nInternalFaces"]
    vf = rng.randn(nC)
    U = rng.randn(nC, 3)
    phi = rng.randn(nF)
    phi[::7] = 0.0            # pos(0) = 1 in the upwind weights
    gamma = 0.5 + rng.rand(nF)
    with tempfile.TemporaryDirectory() as d:
        case = os.path.join(d, "case")
        fv_case.write_case(case, mesh)
        res = fv_case.run_driver(case, mesh, vf, U, phi, gamma)
    l = mesh["owner"][:nF].astype(np.int32)
    u = mesh["neighbour"].astype(np.int32)
    assert np.array_equal(res.pop("owner"), l) and np.array_equal(res.pop("neighbour"), u)
    out = dict(nCells=nC, lowerAddr=l, upperAddr=u, vf=vf, U=U, phi=phi, gamma=gamma)
    out.update({"ref_" + k: (v.astype(np.int32) if k.endswith("_faceCells") else v) for k, v in res.items()})
    return out


GLUE_CASES = {"fvglue_box_6x5x4_cyclic": (6, 5, 4, 5, True), "fvglue_box_3x9x2": (3, 9, 2, 6, False)}


GLUEV_CASES = {"fvglueV_box_5x6x4_cyclic": (5, 6, 4, 11, True)}


def generate_glue(name, mode="glue"):
    nx, ny, nz, seed, cyc = (GLUE_CASES if mode == "glue" else GLUEV_CASES)[name]
    mesh = fv_case.box_mesh(nx, ny, nz, seed=seed, cyclic_x=cyc)
    rng = np.random.RandomState(200 + seed)
    nC, nF = mesh["nCells"], mesh["nInternalFaces"]
    vf, U, phi, gamma = rng.randn(nC), rng.randn(nC, 3), rng.randn(nF), 0.5 + rng.rand(nF)
    with tempfile.TemporaryDirectory() as d:
        case = os.path.join(d, "case")
        fv_case.write_case(case, mesh)
        res = fv_case.run_driver(case, mesh, vf, U, phi, gamma, mode=mode)
    out = dict(nCells=nC, lowerAddr=mesh["owner"][:nF].astype(np.int32), upperAddr=mesh["neighbour"].astype(np.int32))
    for k, v in res.items():
        if k.endswith("_faceCells"):
            v = v.astype(np.int32)
        out[k] = v
    return out


SOLVE_CASES = {"fvsolve_box_14x12x10": (14, 12, 10, 9)}


def generate_solve(name):
    nx, ny, nz, seed = SOLVE_CASES[name]
    mesh = fv_case.box_mesh(nx, ny, nz, seed=seed)
    rng = np.random.RandomState(300 + seed)
    nC, nF = mesh["nCells"], mesh["nInternalFaces"]
    vf, U, phi, gamma = rng.randn(nC), rng.randn(nC, 3), rng.randn(nF), 0.5 + rng.rand(nF)
    with tempfile.TemporaryDirectory() as d:
        case = os.path.join(d, "case")
        fv_case.write_case(case, mesh)
        res = fv_case.run_driver(case, mesh, vf, U, phi, gamma, mode="solve")
    out = dict(nCells=nC, lowerAddr=mesh["owner"][:nF].astype(np.int32), upperAddr=mesh["neighbour"].astype(np.int32))
    for k, v in res.items():
        out[k] = v.astype(np.int32) if k.endswith("_faceCells") else v
    return out


# round 6: a quarter annulus whose x-min / x-max patches are a ROTATIONAL cyclic pair (fv_case.box_mesh sector=True): for the scalar
# field of these solves the transformation of the coupled values is the identity (cyclicLduInterfaceField.C:45-63, rank 0)
SECTOR_CASES = {"fvsolve_sector_8x6x5": (8, 6, 5, 11)}


def generate_sector(name):
    nx, ny, nz, seed = SECTOR_CASES[name]
    mesh = fv_case.box_mesh(nx, ny, nz, seed=seed, cyclic_x=True, sector=True)
    rng = np.random.RandomState(300 + seed)
    nC, nF = mesh["nCells"], mesh["nInternalFaces"]
    vf, U, phi, gamma = rng.randn(nC), rng.randn(nC, 3), rng.randn(nF), 0.5 + rng.rand(nF)
    with tempfile.TemporaryDirectory() as d:
        case = os.path.join(d, "case")
        fv_case.write_case(case, mesh)
        res = fv_case.run_driver(case, mesh, vf, U, phi, gamma, mode="solve", controls="nCellsInCoarsestLevel 10;")
    out = dict(nCells=nC, lowerAddr=mesh["owner"][:nF].astype(np.int32), upperAddr=mesh["neighbour"].astype(np.int32))
    for k, v in res.items():
        out[k] = v.astype(np.int32) if k.endswith("_faceCells") else v
    return out


CHAIN_CASES = {"fvsolve2_halves_6x8x7": (2, 6, 8, 7, 77), "fvsolve4_chain_5x6x6": (4, 5, 6, 6, 78),
               "fvsolve3_chain_asym_5x7x6": (3, 5, 7, 6, 79),
               "fvsolve3_chain_nonblocking_4x7x6": (3, 4, 7, 6, 80)}
# round 6: `directSolveCoarsest true` (GAMGSolver.C:95-106): the coarsest level - all boxes' cells, the cyclic couplings between
# them - through the reference's LUscalarMatrix (LUscalarMatrix.C:128-187); arithmetically the gathered matrix of an N-rank run
# (LUscalarMatrix.C:52-107, :190-318).  "_lu_" in the name switches it on; 8 boxes: a coarsest level of more than 64 cells.
LU_CHAIN_CASES = {"fvsolve4_chain_lu_5x6x6": (4, 5, 6, 6, 84), "fvsolve3_chain_asym_lu_5x7x6": (3, 5, 7, 6, 85)}
LU_GRID_CASES = {"fvsolve8_blocks_lu_2x2x2_4x4x4": ((2, 2, 2), 4, 4, 4, 86, False)}
CHAIN_CASES_ALL = dict(CHAIN_CASES, **LU_CHAIN_CASES)


# 3-D block decompositions and several patches per rank pair (the coarse-interface ordering rule exercised
# against the reference's own cyclicGAMGInterface beyond one patch per pair): name -> ((gx, gy, gz), nx, ny, nz, seed, split)
GRID_CASES = {"fvsolve8_blocks_2x2x2_4x4x4": ((2, 2, 2), 4, 4, 4, 81, False),
              "fvsolve2_split_halves_5x6x6": ((2, 1, 1), 5, 6, 6, 82, True),
              "fvsolve4_blocks_2x2x1_split_4x4x5": ((2, 2, 1), 4, 4, 5, 83, True)}
GRID_CASES_ALL = dict(GRID_CASES, **LU_GRID_CASES)


def generate_chain(name):
    """serial emulation of an N-rank run by the reference itself (see fv_case.chain_box_mesh / grid_box_mesh)"""
    if name in GRID_CASES_ALL:
        g3, nxh, ny, nz, seed, split = GRID_CASES_ALL[name]
        nB = g3[0] * g3[1] * g3[2]
        mesh = fv_case.grid_box_mesh(g3[0], g3[1], g3[2], nxh, ny, nz, split=split)
    else:
        nB, nxh, ny, nz, seed = CHAIN_CASES_ALL[name]
        mesh = fv_case.chain_box_mesh(nB, nxh, ny, nz, axis="z" if "nonblocking" in name else "x")
    rng = np.random.RandomState(seed)
    nC, nF = mesh["nCells"], mesh["nInternalFaces"]
    vf, U, phi, gamma = rng.randn(nC), rng.randn(nC, 3), rng.randn(nF), 0.5 + rng.rand(nF)
    with tempfile.TemporaryDirectory() as d:
        case = os.path.join(d, "case")
        fv_case.write_case(case, mesh)
        res = fv_case.run_driver(case, mesh, vf, U, phi, gamma, mode="solve",
                                 controls="nCellsInCoarsestLevel %d;%s%s%s" % (
                                     10 * nB, " smoother nonBlockingGaussSeidel;" if "nonblocking" in name else "",
                                     " directSolveCoarsest true;" if "_lu_" in name else "",
                                     " asymmetric" if "asym" in name else ""))
    out = dict(nCells=nC, nHalf=mesh["nHalf"], nBoxes=nB, lowerAddr=mesh["owner"][:nF].astype(np.int32),
               upperAddr=mesh["neighbour"].astype(np.int32))
    if "pairs" in mesh:
        out["pairs"] = mesh["pairs"]
    for k, v in res.items():
        out[k] = v.astype(np.int32) if k.endswith("_faceCells") else v
    return out


NONORTH_CASES = {"fvnonorth_box_6x5x4": ("box", (6, 5, 4), 21, False), "fvnonorth_box_5x4x6_cyclic": ("box", (5, 4, 6), 22, True),
                 "fvnonorth_prism_5x4x3": ("prism", (5, 4, 3), 23, False)}


def nonorth_mesh(name):
    kind, dims, seed, cyc = NONORTH_CASES[name]
    if kind == "box":
        return fv_case.box_mesh(*dims, seed=seed, cyclic_x=cyc)
    return fv_case.prism_box_mesh(*dims)


def generate_nonorth(name):
    """a36 / a37 / a39 and the patch halves of a34 / a35 on non-orthogonal meshes (fv_driver mode nonorth): jittered
    graded hex boxes (one with a cyclic pair) and a prism mesh; fixedValue / zeroGradient / cyclic patches"""
    kind, dims, seed, cyc = NONORTH_CASES[name]
    mesh = nonorth_mesh(name)
    rng = np.random.RandomState(400 + seed)
    nC, nF = mesh["nCells"], mesh["nInternalFaces"]
    vf, U, phi, gamma = rng.randn(nC), rng.randn(nC, 3), rng.randn(nF), 0.5 + rng.rand(nF)
    with tempfile.TemporaryDirectory() as d:
        case = os.path.join(d, "case")
        fv_case.write_case(case, mesh)
        res = fv_case.run_driver(case, mesh, vf, U, phi, gamma, mode="nonorth")
    out = dict(nCells=nC, lowerAddr=mesh["owner"][:nF].astype(np.int32), upperAddr=mesh["neighbour"].astype(np.int32))
    for k, v in res.items():
        out[k] = v.astype(np.int32) if k.endswith("_faceCells") else v
    return out


if __name__ == "__main__":
    if not fv_case.driver_available():
        raise SystemExit("oracle/_ref/fv_driver missing: run oracle/build_ref_fv.sh (needs /root/reference)")
    if len(sys.argv) > 1 and sys.argv[1] == "sector":
        for name in SECTOR_CASES:
            data = generate_sector(name)
            np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
            print(name, "cells", data["nCells"], "GAMG perf", data["ref_gamg_perf"], "PCG perf", data["ref_pcg_perf"])
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lu":
        for name in list(LU_CHAIN_CASES) + list(LU_GRID_CASES):
            data = generate_chain(name)
            np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
            print(name, "cells", data["nCells"], "GAMG perf", data["ref_gamg_perf"], "PCG perf", data["ref_pcg_perf"])
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "grid":
        for name in GRID_CASES:
            data = generate_chain(name)
            np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
            print(name, "cells", data["nCells"], "pairs", len(data["pairs"]), "GAMG perf", data["ref_gamg_perf"], "PCG perf", data["ref_pcg_perf"])
        raise SystemExit(0)
    for name in NONORTH_CASES:
        data = generate_nonorth(name)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
        print(name, "cells", data["nCells"], "faces", data["lowerAddr"].size, "patches", int(data["nPatches"][0]),
              "max |corrVec|", float(np.abs(data["nonOrthCorrectionVectors"]).max()))
    if len(sys.argv) > 1 and sys.argv[1] == "nonorth":
        raise SystemExit(0)
    for name in CASES:
        data = generate(name)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
        print(name, "cells", data["nCells"], "faces", data["lowerAddr"].size, sorted(k for k in data if k.startswith("ref_")))
    for name in SOLVE_CASES:
        data = generate_solve(name)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
        print(name, "cells", data["nCells"], "GAMG perf", data["ref_gamg_perf"], "PCG perf", data["ref_pcg_perf"])
    for name in list(CHAIN_CASES_ALL) + list(GRID_CASES_ALL):
        data = generate_chain(name)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
        print(name, "cells", data["nCells"], "GAMG perf", data["ref_gamg_perf"], "PCG perf", data["ref_pcg_perf"])
    for name in SECTOR_CASES:
        data = generate_sector(name)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
        print(name, "cells", data["nCells"], "GAMG perf", data["ref_gamg_perf"], "PCG perf", data["ref_pcg_perf"])
    for name in GLUEV_CASES:
        data = generate_glue(name, mode="glueV")
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
        print(name, "cells", data["nCells"], "vector matrix, patches", int(data["nPatches"][0]))
    for name in GLUE_CASES:
        data = generate_glue(name)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **data)
        print(name, "cells", data["nCells"], "patches", int(data["nPatches"][0]),
              "coupled", [int(data["p%d_coupled" % p][0]) for p in range(int(data["nPatches"][0]))])
