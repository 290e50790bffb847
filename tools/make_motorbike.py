"""BENCH-INPUT PRODUCER (runs HERE, where /root/reference and oracle/_ref/snappyHexMesh exist).

    python tools/make_motorbike.py <name> --q 14 --surface 6 7 [--box-level 4] [--keep-case DIR]

1. writes the simpleFoam motorBike meshing case (oracle/motorbike_case.py) and runs the REFERENCE's own blockMesh and
   snappyHexMesh (castellatedMesh only) on the reference's own motorBike.obj;
2. reads the polyMesh (openfoam-2.2.x_amd/polymesh.py, binary), computes the face / cell geometry with the reference's
   formulas (primitiveMeshFaceCentresAndAreas.C:63-127, primitiveMeshCellCentresAndVols.C:63-170; numpy, vectorised per
   face size) and CHECKS that the mesh is what the compact form below
   assumes: every internal face an axis-aligned square of the finer cell's size, every cell a cube of its cellLevel,
   |Sf| / (n . d) of the real geometry equal to area / normal distance from the levels to 1e-9;
3. stores what the p-equation needs, compressed, under data/motorbike/<name>.npz (git-ignored, travels to the GPU
   box like the rest of oracle/_ref): owner / neighbour of the internal faces (lduAddressing: `ownerCount` per cell and
   `upper`), the face normal direction (0/1/2), cellLevel, the cells of the `outlet` patch, the background cell size -
   ~4 bytes per face.  The 1.5 GB of points / faces stay here.
openfoam-2.2.x_amd/motorbike.py turns the file into the matrix (bench.py --mesh motorbike, tests/test_motorbike.py).
For --small the full polyMesh (points, faces, owner, neighbour, boundary) is stored as well so that the GPU box can run
the product's own device geometry (ldu_mesh_geometry) on a real snappyHexMesh mesh and compare."""
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
import __graft_entry__ as entry  # noqa: E402


def face_geometry(points, faceStart, facePoints):
    """face centres and area vectors: the reference's triangle fan about the average point
    (primitiveMeshFaceCentresAndAreas.C:63-127), vectorised per face size (a castellated mesh has squares with 4 ... 8
    points: hexRef8 puts the mid-edge points of refined neighbours into every face that uses the edge), chunked"""
    nF = faceStart.size - 1
    Cf = np.empty((nF, 3))
    Sf = np.empty((nF, 3))
    size = np.diff(faceStart)
    step = 3000000
    for k in np.unique(size):
        ids = np.flatnonzero(size == k)
        for a in range(0, ids.size, step):
            sel = ids[a:a + step]
            fp = facePoints[faceStart[sel][:, None] + np.arange(k)[None, :]]
            P = points[fp]                                   # [n, k, 3]
            if k == 3:
                Cf[sel] = P.sum(axis=1) / 3.0
                Sf[sel] = 0.5 * np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
                continue
            fCentre = P.sum(axis=1) / float(k)
            sumN = np.zeros_like(fCentre)
            sumA = np.zeros(sel.size)
            sumAc = np.zeros_like(fCentre)
            for pi in range(k):
                p0, p1 = P[:, pi], P[:, (pi + 1) % k]
                c = p0 + p1 + fCentre
                n = np.cross(p1 - p0, fCentre - p0)
                am = np.sqrt((n * n).sum(axis=1))
                sumN += n
                sumA += am
                sumAc += am[:, None] * c
            Cf[sel] = (1.0 / 3.0) * sumAc / sumA[:, None]
            Sf[sel] = 0.5 * sumN
    return Cf, Sf


def cell_geometry(nCells, owner, neighbour, Cf, Sf):
    """primitiveMeshCellCentresAndVols.C:63-170: pyramids about the average of the face centres"""
    nI = neighbour.size
    cEst = np.zeros((nCells, 3))
    nCellFaces = np.zeros(nCells)
    for d in range(3):
        cEst[:, d] = np.bincount(owner, weights=Cf[:, d], minlength=nCells) + np.bincount(neighbour, weights=Cf[:nI, d], minlength=nCells)
    nCellFaces = np.bincount(owner, minlength=nCells) + np.bincount(neighbour, minlength=nCells)
    cEst /= nCellFaces[:, None]
    C = np.zeros((nCells, 3))
    V = np.zeros(nCells)
    pv = (Sf * (Cf - cEst[owner])).sum(axis=1)
    pc = 0.75 * Cf + 0.25 * cEst[owner]
    for d in range(3):
        C[:, d] += np.bincount(owner, weights=pv * pc[:, d], minlength=nCells)
    V += np.bincount(owner, weights=pv, minlength=nCells)
    pv = (Sf[:nI] * (cEst[neighbour] - Cf[:nI])).sum(axis=1)
    pc = 0.75 * Cf[:nI] + 0.25 * cEst[neighbour]
    for d in range(3):
        C[:, d] += np.bincount(neighbour, weights=pv * pc[:, d], minlength=nCells)
    V += np.bincount(neighbour, weights=pv, minlength=nCells)
    C /= V[:, None]
    V *= 1.0 / 3.0
    return C, V


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--q", type=int, default=14)
    ap.add_argument("--box-level", type=int, default=4)
    ap.add_argument("--surface", type=int, nargs=2, default=[6, 7])
    ap.add_argument("--case", default=None, help="use an existing meshed case directory instead of running the generators")
    ap.add_argument("--small", action="store_true", help="also store the whole polyMesh (device-geometry test)")
    ap.add_argument("--only-decomp", action="store_true", help="only (re)write <name>_decomp.npz")
    args = ap.parse_args()
    entry.load_package()
    from openfoam_amd import polymesh
    import motorbike_case as mb
    out_dir = os.path.join(ROOT, "data", "motorbike")
    os.makedirs(out_dir, exist_ok=True)
    case = args.case
    secs = {}
    if case is None:
        case = os.path.join("/tmp", "motorbike_" + args.name)
        shutil.rmtree(case, ignore_errors=True)
        mb.write(case, q=args.q, box_level=args.box_level, surface_levels=tuple(args.surface), max_cells=60000000)
        secs = mb.run(case)
    t0 = time.time()
    m = polymesh.read_polymesh(case)
    nC, nI = m["nCells"], m["nInternalFaces"]
    owner, neighbour = m["owner"], m["neighbour"]
    l, u = polymesh.ldu_addressing(m)
    level, _ = polymesh.read_labels(os.path.join(case, "constant", "polyMesh", "cellLevel"))
    if level.size != nC:
        raise SystemExit("cellLevel has %d entries for %d cells" % (level.size, nC))
    Cf, Sf = face_geometry(m["points"], m["faceStart"], m["facePoints"])
    C, V = cell_geometry(nC, owner, neighbour, Cf, Sf)
    h0 = 20.0 / (5 * args.q)
    h = h0 / (1 << level.astype(np.int64))
    # checks: cubes, axis-aligned square faces of the finer cell, coefficient from the levels
    relV = np.abs(V / h ** 3 - 1.0).max()
    magSf = np.sqrt((Sf * Sf).sum(axis=1))
    dirs = np.argmax(np.abs(Sf[:nI]), axis=1).astype(np.uint8)
    offaxis = (1.0 - np.abs(Sf[np.arange(nI), dirs]) / magSf[:nI]).max()
    lf = np.maximum(level[l], level[u]).astype(np.int64)
    hf = h0 / (1 << lf)
    relA = np.abs(magSf[:nI] / (hf * hf) - 1.0).max()
    d = C[u] - C[l]
    nd = np.abs((d * Sf[:nI]).sum(axis=1)) / magSf[:nI]
    dn = np.where(level[l] != level[u], 1.5 * hf, hf)
    relD = np.abs(nd / dn - 1.0).max()
    if max(relV, offaxis, relA, relD) > 1e-9 or np.abs(level[l].astype(int) - level[u]).max() > 1:
        raise SystemExit("the mesh is not the castellated octree the compact form assumes: "
                         "V %.2e axis %.2e area %.2e delta %.2e" % (relV, offaxis, relA, relD))
    outlet = [p for p in m["patches"] if p["name"] == "outlet"][0]
    out_faces = np.arange(outlet["startFace"], outlet["startFace"] + outlet["nFaces"])
    out_cells = owner[out_faces].astype(np.int32)
    out_dn = np.abs(((Cf[out_faces] - C[out_cells]) * Sf[out_faces]).sum(axis=1)) / magSf[out_faces]
    if np.abs(out_dn / (0.5 * h[out_cells]) - 1.0).max() > 1e-9:
        raise SystemExit("outlet faces are not half a cell from their cell centres")
    ownerCount = np.bincount(l, minlength=nC).astype(np.uint8)
    if np.bincount(l, minlength=nC).max() > 255:
        raise SystemExit("more than 255 owned faces in a cell")
    meta = dict(name=args.name, q=args.q, box_level=args.box_level, surface_levels=list(args.surface), h0=h0, nCells=int(nC),
                nInternalFaces=int(nI), nFaces=int(owner.size), nPoints=int(m["points"].shape[0]),
                cells_per_level=np.bincount(level).tolist(), generator_seconds=secs,
                checks=dict(volume=float(relV), off_axis=float(offaxis), area=float(relA), normal_distance=float(relD)),
                patches=[(p["name"], p["type"], p["nFaces"]) for p in m["patches"] if p["nFaces"]][:12],
                source="the reference's blockMesh + snappyHexMesh (castellatedMesh only) on tutorials/resources/geometry/"
                       "motorBike.obj.gz; dictionaries: oracle/motorbike_case.py")
    # the reference's own hierarchical decomposition of the cell centres (libdecompositionMethods through oracle/decomp_driver.C;
    # motorBike/system/decomposeParDict:17-33: method hierarchical, n (3 2 1), delta 0.001, order xyz) for 2 / 4 / 6 / 8 ranks
    import subprocess
    drv = os.path.join(ROOT, "oracle", "_ref", "decomp_driver")
    if os.path.exists(drv):
        cfile = os.path.join("/tmp", "centres_%s.bin" % args.name)
        np.ascontiguousarray(C).tofile(cfile)
        dec = {}
        for nr, nn in ((2, "2 1 1"), (4, "2 2 1"), (6, "3 2 1"), (8, "2 2 2")):
            ofile = cfile + ".proc%d" % nr
            r = subprocess.run([drv, cfile, str(nC), ofile, "numberOfSubdomains %d; method hierarchical; hierarchicalCoeffs { n (%s); "
                                "delta 0.001; order xyz; }" % (nr, nn)], env=mb.env(), capture_output=True, text=True)
            if r.returncode:
                raise SystemExit("decomp_driver failed:\n" + r.stdout[-1000:] + r.stderr[-1000:])
            dec["proc%d" % nr] = np.fromfile(ofile, dtype=np.int32).astype(np.uint8)
            print(r.stdout.strip().splitlines()[-1][:200])
            os.remove(ofile)
        os.remove(cfile)
        np.savez_compressed(os.path.join(out_dir, args.name + "_decomp.npz"), **dec)
    if args.only_decomp:
        return
    path = os.path.join(out_dir, args.name + ".npz")
    np.savez_compressed(path, ownerCount=ownerCount, upper=u.astype(np.int32), dirs=dirs, cellLevel=level.astype(np.uint8),
                        outletCells=out_cells, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print("wrote %s (%.1f MB) in %.0f s" % (path, os.path.getsize(path) / 1e6, time.time() - t0))
    print(json.dumps(meta))
    if args.small:
        p2 = os.path.join(out_dir, args.name + "_polymesh.npz")
        np.savez_compressed(p2, points=m["points"], faceStart=m["faceStart"], facePoints=m["facePoints"], owner=owner, neighbour=neighbour,
                            patchStart=np.array([p["startFace"] for p in m["patches"]], dtype=np.int32),
                            patchSize=np.array([p["nFaces"] for p in m["patches"]], dtype=np.int32),
                            V=V, magSf=magSf)
        print("wrote %s (%.1f MB)" % (p2, os.path.getsize(p2) / 1e6))


if __name__ == "__main__":
    main()
