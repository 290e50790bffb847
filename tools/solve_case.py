"""Run the GAMG p-solve of bench.py on a REAL case's mesh: reads constant/polyMesh (ascii) of an OpenFOAM-2.2.x
case, builds the geometry and the fvm::laplacian coefficients on the device, optionally renumbers the cells with
Foam::bandCompression (what renumberMesh does), and reports V-cycles/s like bench.py.

    python tools/solve_case.py <case> [--renumber] [--steps 5]

(The genuine motorBike mesh needs blockMesh/snappyHexMesh from a full OpenFOAM build; any case whose mesh was
written with `writeFormat ascii; writeCompression off;` works.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--renumber", action="store_true")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    entry.load_package()
    from openfoam_amd import capi, polymesh
    t0 = time.perf_counter()
    r = polymesh.read_polymesh(args.case)
    t_read = time.perf_counter() - t0
    ctx = capi.Context(0)
    nC, nI = r["nCells"], r["nInternalFaces"]
    Cf, Sf, C, V = capi.mesh_geometry(ctx, r["points"], r["faceStart"], r["facePoints"], r["owner"], r["neighbour"], nC)
    w, delta, magSf = capi.mesh_interpolation_factors(ctx, r["owner"], r["neighbour"], Cf, Sf, C)
    l, u = r["owner"][:nI].astype(np.int32), r["neighbour"].astype(np.int32)
    # faceAreaPairGAMGAgglomeration.C:56-68: |Sf/sqrt(magSf) cmptMultiplied by (1 1.01 1.02)|
    s = Sf[:nI] / np.sqrt(magSf)[:, None] * np.array([1.0, 1.01, 1.02])
    fw = np.sqrt(s[:, 0] * s[:, 0] + s[:, 1] * s[:, 1] + s[:, 2] * s[:, 2])
    coef = delta * magSf
    if args.renumber:
        order = capi.band_compression(nC, l, u)
        l, u, fmap, _ = capi.renumber_addressing(nC, l, u, order)
        coef, fw, V = coef[fmap], fw[fmap], V[order]
    else:
        l, u = polymesh.ldu_addressing(r)
    a = capi.Addressing(ctx, nC, l, u, fw)
    diag = np.zeros(nC)
    np.add.at(diag, l, coef)
    np.add.at(diag, u, coef)
    diag += 1e-3 * V / V.mean() * np.mean(coef)     # pins the level like a reference cell
    m = capi.Matrix(a)
    src = np.random.RandomState(0).randn(nC)
    kw = dict(solver="GAMG", smoother="GaussSeidel", agglomerator="faceAreaPair", nCellsInCoarsestLevel=10,
              mergeLevels=1, cacheAgglomeration=True, tolerance=1e-6, relTol=0.01, maxIter=50)
    secs, its = [], []
    for rep in range(args.steps + 1):
        m.set_coeffs(diag, -coef)
        x, perf = m.solve(np.zeros(nC), src, **kw)
        if rep:
            secs.append(perf["solveSeconds"]); its.append(perf["nIterations"])
    info = a.info()
    print(json.dumps({"case": os.path.abspath(args.case), "nCells": nC, "nInternalFaces": nI, "renumbered": args.renumber,
                      "dependency_levels": int(info["nLevels"]), "V_cycles_per_s": sum(its) / sum(secs),
                      "ms_per_solve": 1e3 * sum(secs) / len(secs), "V_cycles_per_solve": its[-1],
                      "read_s": round(t_read, 3), "sweep_engine_finest": a.sweep_engine(2)}))


if __name__ == "__main__":
    main()
