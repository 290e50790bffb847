"""TEST INFRASTRUCTURE - writes BASELINE config C2, simpleFoam on the pitzDaily backward-facing step, as a case directory.

Mesh: the arrays of tests/golden/pitzDaily_12225.npz - what the REFERENCE's own blockMesh library made of the tutorial's
blockMeshDict (tests/golden/make_pitzdaily_golden.py; 12 225 cells, 24 170 internal faces) - written back as
constant/polyMesh/{points,faces,owner,neighbour,boundary}.  Dictionaries and fields: the numbers of the tutorial
(tutorials/incompressible/simpleFoam/pitzDaily: system/fvSchemes, system/fvSolution, constant/RASProperties,
constant/transportProperties, 0/{U,p,k,epsilon,nut}) restated as our own text, so that the case exists where
/root/reference does not (the GPU box).  kEpsilon, bounded Gauss upwind, Gauss linear corrected, p relaxed 0.3, U / k /
epsilon 0.7; p: the tutorial's PCG + DIC or the motorBike GAMG block."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
GOLDEN = os.path.join(HERE, "..", "tests", "golden", "pitzDaily_12225.npz")

HEAD = "FoamFile\n{\n    version     2.0;\n    format      ascii;\n    class       %s;\n    object      %s;\n}\n"

GAMG = ("        solver          GAMG;\n        tolerance       1e-06;\n        relTol          0.01;\n"
        "        smoother        GaussSeidel;\n        nPreSweeps      0;\n        nPostSweeps     2;\n"
        "        cacheAgglomeration on;\n        agglomerator    faceAreaPair;\n        nCellsInCoarsestLevel 10;\n"
        "        mergeLevels     1;")
PCG = "        solver          PCG;\n        preconditioner  DIC;\n        tolerance       1e-06;\n        relTol          0.01;"


def env():
    return dict(os.environ, WM_PROJECT="OpenFOAM", WM_PROJECT_VERSION="2.2.x", WM_PROJECT_DIR=REF,
                LD_LIBRARY_PATH=REF + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FOAM_SIGFPE="false")


def available():
    return os.path.exists(os.path.join(REF, "simpleFoam")) and os.path.exists(GOLDEN)


def _w(path, cls, obj, body):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(HEAD % (cls, obj) + body)


def write(case, steps=30, libs=None, p_solver=None, gauss="Gauss"):
    """gauss: the name the gradient / convection / laplacian schemes are selected under - "Gauss" (the reference's classes) or
    "hipGauss" (libhipFvSchemes.so: the same schemes with their face loops on the device; needs that library in `libs`)"""
    g = np.load(GOLDEN, allow_pickle=True)
    pm = os.path.join(case, "constant", "polyMesh")
    pts, fs, fp = g["points"], g["faceStart"], g["facePoints"]
    _w(os.path.join(pm, "points"), "vectorField", "points",
       "%d\n(\n%s\n)\n" % (len(pts), "\n".join("(%.17g %.17g %.17g)" % tuple(p) for p in pts)))
    _w(os.path.join(pm, "faces"), "faceList", "faces",
       "%d\n(\n%s\n)\n" % (len(fs) - 1, "\n".join("%d(%s)" % (fs[i + 1] - fs[i], " ".join(str(int(v)) for v in fp[fs[i]:fs[i + 1]]))
                                                      for i in range(len(fs) - 1))))
    for name in ("owner", "neighbour"):
        _w(os.path.join(pm, name), "labelList", name, "%d\n(\n%s\n)\n" % (len(g[name]), "\n".join(str(int(v)) for v in g[name])))
    _w(os.path.join(pm, "boundary"), "polyBoundaryMesh", "boundary",
       "%d\n(\n%s)\n" % (len(g["patchNames"]), "".join("%s\n{\n    type %s;\n    nFaces %d;\n    startFace %d;\n}\n" % (n, t, s, st)
                                                       for n, t, s, st in zip(g["patchNames"], g["patchTypes"], g["patchSize"], g["patchStart"]))))
    _w(os.path.join(case, "system", "controlDict"), "dictionary", "controlDict",
       "application simpleFoam;\nstartFrom startTime;\nstartTime 0;\nstopAt endTime;\nendTime %d;\ndeltaT 1;\n"
       "writeControl timeStep;\nwriteInterval 100000;\npurgeWrite 0;\nwriteFormat ascii;\nwritePrecision 6;\n"
       "writeCompression off;\ntimeFormat general;\ntimePrecision 6;\nrunTimeModifiable false;\n%s\n"
       % (steps, ("libs (%s);" % " ".join('"%s"' % l for l in libs)) if libs else ""))
    _w(os.path.join(case, "system", "fvSchemes"), "dictionary", "fvSchemes", """
ddtSchemes { default steadyState; }
gradSchemes { default GAUSS linear; grad(p) GAUSS linear; grad(U) GAUSS linear; }
divSchemes
{
    default none;
    div(phi,U) bounded GAUSS upwind;
    div(phi,k) bounded GAUSS upwind;
    div(phi,epsilon) bounded GAUSS upwind;
    div((nuEff*dev(T(grad(U))))) Gauss linear;
}
laplacianSchemes
{
    default none;
    laplacian(nuEff,U) GAUSS linear corrected;
    laplacian((1|A(U)),p) GAUSS linear corrected;
    laplacian(DkEff,k) GAUSS linear corrected;
    laplacian(DepsilonEff,epsilon) GAUSS linear corrected;
}
interpolationSchemes { default linear; interpolate(U) linear; }
snGradSchemes { default corrected; }
fluxRequired { default no; p ; }
""".replace("GAUSS", gauss))
    bicg = "{ solver PBiCG; preconditioner DILU; tolerance 1e-05; relTol 0.1; }"
    _w(os.path.join(case, "system", "fvSolution"), "dictionary", "fvSolution", """
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
""" % (p_solver or PCG, bicg, bicg, bicg))
    _w(os.path.join(case, "constant", "RASProperties"), "dictionary", "RASProperties",
       "\nRASModel kEpsilon;\nturbulence on;\nprintCoeffs on;\n")
    _w(os.path.join(case, "constant", "transportProperties"), "dictionary", "transportProperties",
       "\ntransportModel Newtonian;\nnu nu [ 0 2 -1 0 0 0 0 ] 1e-05;\n")

    def field(name, cls, dims, internal, inlet, outlet, walls):
        _w(os.path.join(case, "0", name), cls, name,
           "\ndimensions %s;\ninternalField uniform %s;\nboundaryField\n{\n    inlet { %s }\n    outlet { %s }\n"
           "    upperWall { %s }\n    lowerWall { %s }\n    frontAndBack { type empty; }\n}\n"
           % (dims, internal, inlet, outlet, walls, walls))
    field("U", "volVectorField", "[0 1 -1 0 0 0 0]", "(0 0 0)", "type fixedValue; value uniform (10 0 0);",
          "type zeroGradient;", "type fixedValue; value uniform (0 0 0);")
    field("p", "volScalarField", "[0 2 -2 0 0 0 0]", "0", "type zeroGradient;", "type fixedValue; value uniform 0;",
          "type zeroGradient;")
    field("k", "volScalarField", "[0 2 -2 0 0 0 0]", "0.375", "type fixedValue; value uniform 0.375;", "type zeroGradient;",
          "type kqRWallFunction; value uniform 0.375;")
    field("epsilon", "volScalarField", "[0 2 -3 0 0 0 0]", "14.855", "type fixedValue; value uniform 14.855;",
          "type zeroGradient;", "type epsilonWallFunction; value uniform 14.855;")
    field("nut", "volScalarField", "[0 2 -1 0 0 0 0]", "0", "type calculated; value uniform 0;",
          "type calculated; value uniform 0;", "type nutkWallFunction; value uniform 0;")


def run(case, extra_env=None):
    e = env()
    if extra_env:
        e.update(extra_env)
    r = subprocess.run([os.path.join(REF, "simpleFoam"), "-case", case], env=e, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("simpleFoam failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    return r.stdout + r.stderr
