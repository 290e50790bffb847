"""BENCH-INPUT PRODUCER (runs HERE, where /root/reference and oracle/_ref exist): the p-matrix of a REAL SIMPLE
iteration of the REFERENCE's simpleFoam on a SNAPPED + LAYERED motorBike mesh.

    python tools/make_motorbike_matrix.py <name> --q 8 [--surface 5 6] [--iteration 3] [--case DIR]

1. the tutorial's meshing case with `snap true; addLayers true;` (motorBike/system/snappyHexMeshDict:18-20; one layer on
   "(lowerWall|motorBike).*", :176-182) through the reference's own blockMesh + snappyHexMesh (oracle/motorbike_case.py) -
   polyhedral snapped cells, layer prisms, non-orthogonal faces; or an already meshed case (--case);
2. the flow case of oracle/motorbike_simplefoam_case.py on that polyMesh, `solver dumpGAMG` for p (oracle/dump_solver.C: the
   reference's GAMGSolver behind a matrix dump), --iteration SIMPLE iterations of oracle/_ref/simpleFoam;
3. the dumped matrix of the last p-solve - `laplacian((1|A(U)),p)` with the fixedValue outlet's boundary coefficients and the
   non-orthogonal correction in its source - stored under data/motorbike/<name>.npz (git-ignored, travels to the GPU box):
   lduAddressing, diag / upper (f64), source (f64), the faceAreaPair weights of the mesh's own face area vectors
   (faceAreaPairGAMGAgglomeration.C:59-72; f32), mesh statistics.
openfoam-2.2.x_amd/motorbike.py: dumped_problem(name) turns it into a problem dict (tests/test_motorbike.py, bench.py)."""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--q", type=int, default=8)
    ap.add_argument("--surface", type=int, nargs=2, default=[5, 6])
    ap.add_argument("--iteration", type=int, default=3)
    ap.add_argument("--case", default=None, help="an already meshed case directory (snappyHexMesh -overwrite)")
    ap.add_argument("--no-layers", action="store_true")
    args = ap.parse_args()
    import motorbike_case as mb
    import motorbike_simplefoam_case as mc
    from test_simplefoam_motorbike import read_dump
    secs = {}
    mesh_case = args.case
    if mesh_case is None:
        mesh_case = os.path.join("/tmp", "motorbike_" + args.name)
        shutil.rmtree(mesh_case, ignore_errors=True)
        mb.write(mesh_case, q=args.q, box_level=4, surface_levels=tuple(args.surface), max_cells=60000000, snap=True,
                 layers=not args.no_layers)
        secs = mb.run(mesh_case)
    log_mesh = open(os.path.join(mesh_case, "log.snappyHexMesh")).read()
    stages = [ln.strip() for ln in log_mesh.splitlines() if ln.startswith(("Snapped mesh :", "Layer mesh :", "Refined mesh :"))]
    flow = os.path.join("/tmp", "simplefoam_" + args.name)
    shutil.rmtree(flow, ignore_errors=True)
    dump = os.path.join(flow, "p_matrix.bin")
    mc.write(flow, steps=args.iteration, libs=[os.path.join(ROOT, "oracle", "_ref", "libdumpSolver.so")], mesh_from=mesh_case,
             p_solver="dumpGAMG")
    t0 = time.time()
    log = mc.run(flow, extra_env={"LDU_DUMP_MATRIX": "p:%d:%s" % (args.iteration, dump)})
    secs["simpleFoam"] = time.time() - t0
    p_lines = [ln for ln in log.splitlines() if "Solving for p" in ln]
    p = read_dump(dump)
    nC, l, u = p["nCells"], p["lowerAddr"], p["upperAddr"]
    Sf = p["Sf"]
    magSf = np.sqrt((Sf * Sf).sum(axis=1))
    comp = (Sf / np.sqrt(magSf)[:, None]) * np.array([1.0, 1.01, 1.02])[None, :]
    w = np.sqrt((comp * comp).sum(axis=1))
    # mesh statistics: internal faces per cell (hex 6 minus boundary faces; snapped polyhedra and split hexes more), and the
    # non-orthogonality the matrix carries: the spread of upper / (|Sf| ... ) is not recoverable without the geometry, so the
    # reference's own checkMesh figures are taken from the snappyHexMesh log
    deg = np.bincount(l, minlength=nC) + np.bincount(u, minlength=nC)
    nonortho = [ln.strip() for ln in log_mesh.splitlines() if "non-orthogonality" in ln][-1:]
    assert np.all(l < u) and np.all(np.diff(l.astype(np.int64) * nC + u) > 0)
    own = np.bincount(l, minlength=nC)
    if own.max() > 255:
        raise SystemExit("more than 255 owned faces in a cell")
    meta = dict(name=args.name, nCells=int(nC), nInternalFaces=int(l.size), q=args.q, surface_levels=list(args.surface),
                snap=True, layers=not args.no_layers, mesh_stages=stages, iteration=args.iteration, solver_lines_p=p_lines,
                internal_faces_per_cell=np.bincount(deg).tolist(), seconds=secs, checkmesh=nonortho,
                symmetric="lower" not in p,
                source="the reference's blockMesh + snappyHexMesh (castellate, snap, addLayers) on motorBike.obj, then %d SIMPLE "
                       "iterations of the reference's simpleFoam (oracle/motorbike_simplefoam_case.py): the matrix its last p-solve "
                       "was handed (oracle/dump_solver.C)" % args.iteration)
    out = os.path.join(ROOT, "data", "motorbike", args.name + ".npz")
    np.savez_compressed(out, ownerCount=own.astype(np.uint8), upperAddr=u.astype(np.int32), diag=p["diag"], upper=p["upper"],
                        source=p["source"], faceWeights=w.astype(np.float32), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print("wrote %s (%.1f MB)" % (out, os.path.getsize(out) / 1e6))
    print(json.dumps(meta, indent=1)[:3000])


if __name__ == "__main__":
    main()
