"""TEST INFRASTRUCTURE - writes BASELINE config C3's flow case at the tutorial's own mesh size: simpleFoam on the mesh the
REFERENCE's blockMesh + snappyHexMesh made of the reference's motorBike.obj (data/motorbike/mbtut_polymesh.npz, written by
tools/make_motorbike.py --small: ~321 k cells, ~961 k internal faces, 72 patches, refinement levels 0..6; see mesh_identity).

Dictionaries and fields restate the numbers of tutorials/incompressible/simpleFoam/motorBike (system/fvSolution: p GAMG /
GaussSeidel / nPreSweeps 0 / nPostSweeps 2 / faceAreaPair / nCellsInCoarsestLevel 10 / mergeLevels 1, tolerance 1e-7,
relTol 0.1; U and the turbulence fields smoothSolver + GaussSeidel, nSweeps 1, relTol 0.1; relaxation p 0.3, equations 0.7;
0/: inlet 20 m/s, moving ground 20 m/s, nu 1.5e-5, k 0.24, omega 1.78) as our own text, so that the case exists where
/root/reference does not (the GPU box).  What differs from the tutorial, and why - oracle/_ref/simpleFoam is linked from
the units pitzDaily (config C2) reaches and nothing else (oracle/build_ref_fv.sh; no stand-ins):
  * kEpsilon instead of kOmegaSST (epsilon = Cmu k omega = 0.0384), upwind instead of linearUpwindV for div(phi,U);
  * the far-field sides (upperWall, frontAndBack) are zeroGradient instead of slip, the outlet zeroGradient instead of
    inletOutlet - those patch-field classes are not in the binary.
None of this changes what the test is for: the p-equation the application assembles on the real castellated mesh
(rAU = 1/UEqn().A() under `laplacian((1|A(U)),p)`), solved by the GAMG block of the motorBike tutorial."""
import os

import numpy as np

import pitzdaily_case as pz

HERE = os.path.dirname(os.path.abspath(__file__))
STORE = os.path.join(HERE, "..", "data", "motorbike", "mbtut_polymesh.npz")
BOX_PATCHES = ("frontAndBack", "inlet", "outlet", "lowerWall", "upperWall")     # blockMeshDict order; the rest: the surface's regions

GAMG = ("        solver          GAMG;\n        tolerance       1e-07;\n        relTol          0.1;\n"
        "        smoother        GaussSeidel;\n        nPreSweeps      0;\n        nPostSweeps     2;\n"
        "        cacheAgglomeration on;\n        agglomerator    faceAreaPair;\n        nCellsInCoarsestLevel 10;\n"
        "        mergeLevels     1;")
env = pz.env
run = pz.run


def available():
    return os.path.exists(os.path.join(pz.REF, "simpleFoam")) and os.path.exists(STORE)


def mesh_identity():
    """which mesh the store holds.  snappyHexMesh is NOT reproducible across hosts (the same binaries and dictionaries gave
    321 362 cells on one machine of the pool and 321 348 on another: borderline inside / outside decisions follow the host's
    libm variants), so a log fixture made on the mesh is tied to it: the fixture records this identity and the tests compare
    before they compare anything else."""
    import hashlib
    g = np.load(STORE)
    nb = np.ascontiguousarray(g["neighbour"], dtype=np.int32)
    return dict(nCells=int(g["owner"].max()) + 1, nInternalFaces=int(nb.size), neighbour_sha1=hashlib.sha1(nb.tobytes()).hexdigest())


def _list(path, cls, obj, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(pz.HEAD % (cls, obj) + "%d\n(\n" % len(rows))
        f.write("\n".join(rows))
        f.write("\n)\n")


def write(case, steps=3, libs=None, gauss="Gauss", smooth="smoothSolver", mesh_from=None, p_solver="GAMG"):
    """mesh_from: a meshed case directory (snappyHexMesh -overwrite) whose polyMesh is used as it is - e.g. the snapped + layered
    meshes of tools/make_motorbike_matrix.py - instead of the stored tutorial-size mesh; p_solver: the name under `solver` for p
    (dumpGAMG = oracle/dump_solver.C: the reference's GAMGSolver behind a matrix dump)"""
    pm = os.path.join(case, "constant", "polyMesh")
    if mesh_from is not None:
        import shutil
        os.makedirs(pm, exist_ok=True)
        for f in ("points", "faces", "owner", "neighbour", "boundary"):
            shutil.copy(os.path.join(mesh_from, "constant", "polyMesh", f), os.path.join(pm, f))
        _write_dicts_and_fields(case, steps, libs, gauss, smooth, p_solver)
        return
    g = np.load(STORE)
    pts, fs, fp = g["points"], g["faceStart"], g["facePoints"]
    _list(os.path.join(pm, "points"), "vectorField", "points", ["(%.17g %.17g %.17g)" % (p[0], p[1], p[2]) for p in pts.tolist()])
    fpl, fsl = fp.tolist(), fs.tolist()
    _list(os.path.join(pm, "faces"), "faceList", "faces",
          ["%d(%s)" % (fsl[i + 1] - fsl[i], " ".join(map(str, fpl[fsl[i]:fsl[i + 1]]))) for i in range(len(fsl) - 1)])
    for name in ("owner", "neighbour"):
        _list(os.path.join(pm, name), "labelList", name, [str(v) for v in g[name].tolist()])
    sizes, starts = g["patchSize"].tolist(), g["patchStart"].tolist()
    names = list(BOX_PATCHES) + ["motorBike_%d" % i for i in range(len(sizes) - len(BOX_PATCHES))]
    types = ["wall" if (n == "lowerWall" or n.startswith("motorBike_")) else "patch" for n in names]
    pz._w(os.path.join(pm, "boundary"), "polyBoundaryMesh", "boundary",
          "%d\n(\n%s)\n" % (len(names), "".join("%s\n{\n    type %s;\n    nFaces %d;\n    startFace %d;\n}\n" % r
                                                 for r in zip(names, types, sizes, starts))))
    _write_dicts_and_fields(case, steps, libs, gauss, smooth, p_solver)


def _write_dicts_and_fields(case, steps, libs, gauss, smooth, p_solver="GAMG"):
    pz._w(os.path.join(case, "system", "controlDict"), "dictionary", "controlDict",
          "application simpleFoam;\nstartFrom startTime;\nstartTime 0;\nstopAt endTime;\nendTime %d;\ndeltaT 1;\n"
          "writeControl timeStep;\nwriteInterval 100000;\npurgeWrite 0;\nwriteFormat ascii;\nwritePrecision 6;\n"
          "writeCompression off;\ntimeFormat general;\ntimePrecision 6;\nrunTimeModifiable false;\n%s\n"
          % (steps, ("libs (%s);" % " ".join('"%s"' % l for l in libs)) if libs else ""))
    pz._w(os.path.join(case, "system", "fvSchemes"), "dictionary", "fvSchemes", """
ddtSchemes { default steadyState; }
gradSchemes { default GAUSS linear; }
divSchemes
{
    default none;
    div(phi,U) bounded GAUSS upwind;
    div(phi,k) bounded GAUSS upwind;
    div(phi,epsilon) bounded GAUSS upwind;
    div((nuEff*dev(T(grad(U))))) Gauss linear;
}
laplacianSchemes { default GAUSS linear corrected; }
interpolationSchemes { default linear; }
snGradSchemes { default corrected; }
fluxRequired { default no; p ; }
""".replace("GAUSS", gauss))
    if smooth == "smoothSolver":
        other = "{ solver smoothSolver; smoother GaussSeidel; tolerance 1e-08; relTol 0.1; nSweeps 1; }"
    else:
        other = "{ solver PBiCG; preconditioner DILU; tolerance 1e-08; relTol 0.1; }"
    pz._w(os.path.join(case, "system", "fvSolution"), "dictionary", "fvSolution", """
solvers
{
    p
    {
%s
    }
    U %s
    k %s
    epsilon %s
}
SIMPLE { nNonOrthogonalCorrectors 0; }
relaxationFactors
{
    fields { p 0.3; }
    equations { U 0.7; k 0.7; epsilon 0.7; }
}
""" % (GAMG.replace("solver          GAMG;", "solver          %s;" % p_solver), other, other, other))
    pz._w(os.path.join(case, "constant", "RASProperties"), "dictionary", "RASProperties",
          "\nRASModel kEpsilon;\nturbulence on;\nprintCoeffs on;\n")
    pz._w(os.path.join(case, "constant", "transportProperties"), "dictionary", "transportProperties",
          "\ntransportModel Newtonian;\nnu nu [ 0 2 -1 0 0 0 0 ] 1.5e-05;\n")

    def field(name, cls, dims, internal, inlet, outlet, ground, bike, sides):
        pz._w(os.path.join(case, "0", name), cls, name,
              "\ndimensions %s;\ninternalField uniform %s;\nboundaryField\n{\n    inlet { %s }\n    outlet { %s }\n"
              "    lowerWall { %s }\n    \"motorBike_.*\" { %s }\n    upperWall { %s }\n    frontAndBack { %s }\n}\n"
              % (dims, internal, inlet, outlet, ground, bike, sides, sides))
    zg = "type zeroGradient;"
    field("U", "volVectorField", "[0 1 -1 0 0 0 0]", "(20 0 0)", "type fixedValue; value uniform (20 0 0);", zg,
          "type fixedValue; value uniform (20 0 0);", "type fixedValue; value uniform (0 0 0);", zg)
    field("p", "volScalarField", "[0 2 -2 0 0 0 0]", "0", zg, "type fixedValue; value uniform 0;", zg, zg, zg)
    field("k", "volScalarField", "[0 2 -2 0 0 0 0]", "0.24", "type fixedValue; value uniform 0.24;", zg,
          "type kqRWallFunction; value uniform 0.24;", "type kqRWallFunction; value uniform 0.24;", zg)
    field("epsilon", "volScalarField", "[0 2 -3 0 0 0 0]", "0.0384", "type fixedValue; value uniform 0.0384;", zg,
          "type epsilonWallFunction; value uniform 0.0384;", "type epsilonWallFunction; value uniform 0.0384;", zg)
    calc = "type calculated; value uniform 0;"
    field("nut", "volScalarField", "[0 2 -1 0 0 0 0]", "0", calc, calc,
          "type nutkWallFunction; value uniform 0;", "type nutkWallFunction; value uniform 0;", calc)
