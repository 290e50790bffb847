"""TEST INFRASTRUCTURE - writes BASELINE config C5's application, interFoam on the damBreak tutorial, as a case directory.

Mesh and initial phase fraction: tests/golden/damBreak_2268.npz - what the REFERENCE's own blockMesh made of the tutorial's
blockMeshDict (five blocks, 2 268 cells) and the reference's own setFields of the tutorial's setFieldsDict (water column
0 <= x <= 0.1461, 0 <= y <= 0.292) - written back as constant/polyMesh/* and 0/alpha1 (tests/golden/make_dambreak_golden.py).
Dictionaries and fields: the numbers of tutorials/multiphase/interFoam/laminar/damBreak (system/fvSchemes, system/fvSolution,
constant/transportProperties, constant/g, constant/turbulenceProperties, 0/U, 0/p_rgh, 0/alpha1.org) restated as our own text so
that the case exists where /root/reference does not (the GPU box): water / air (rho 1000 / 1, nu 1e-6 / 1.48e-5, sigma 0.07),
laminar, PIMPLE without momentum predictor, 3 correctors, MULES with interface compression, adjustable time step (maxCo 0.5).
p_rgh: the tutorial's PCG + DIC (tolerance 1e-7, relTol 0.05; p_rghFinal relTol 0) or, as BASELINE.json words C5, GAMG."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
GOLDEN = os.path.join(HERE, "..", "tests", "golden", "damBreak_2268.npz")

HEAD = "FoamFile\n{\n    version     2.0;\n    format      ascii;\n    class       %s;\n    object      %s;\n}\n"

PCG = "solver PCG; preconditioner DIC; tolerance 1e-07; relTol 0.05;"
GAMG = ("solver GAMG; tolerance 1e-07; relTol 0.05; smoother GaussSeidel; nPreSweeps 0; nPostSweeps 2; cacheAgglomeration on; "
        "agglomerator faceAreaPair; nCellsInCoarsestLevel 10; mergeLevels 1;")


def env():
    return dict(os.environ, WM_PROJECT="OpenFOAM", WM_PROJECT_VERSION="2.2.x", WM_PROJECT_DIR=REF,
                LD_LIBRARY_PATH=REF + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FOAM_SIGFPE="false")


def available():
    return os.path.exists(os.path.join(REF, "interFoam")) and os.path.exists(GOLDEN)


def _w(path, cls, obj, body):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(HEAD % (cls, obj) + body)


def write_dictionaries(case, end_time=0.02, libs=None, p_solver=None):
    _w(os.path.join(case, "system", "controlDict"), "dictionary", "controlDict",
       "application interFoam;\nstartFrom startTime;\nstartTime 0;\nstopAt endTime;\nendTime %.10g;\ndeltaT 0.001;\n"
       "writeControl adjustableRunTime;\nwriteInterval 1000;\npurgeWrite 0;\nwriteFormat ascii;\nwritePrecision 6;\n"
       "writeCompression uncompressed;\ntimeFormat general;\ntimePrecision 6;\nrunTimeModifiable no;\nadjustTimeStep yes;\n"
       "maxCo 0.5;\nmaxAlphaCo 0.5;\nmaxDeltaT 1;\n%s\n"
       % (end_time, ("libs (%s);" % " ".join('"%s"' % l for l in libs)) if libs else ""))
    _w(os.path.join(case, "system", "fvSchemes"), "dictionary", "fvSchemes", """
ddtSchemes { default Euler; }
gradSchemes { default Gauss linear; }
divSchemes
{
    div(rho*phi,U)  Gauss limitedLinearV 1;
    div(phi,alpha)  Gauss vanLeer;
    div(phirb,alpha) Gauss interfaceCompression;
    div((muEff*dev(T(grad(U))))) Gauss linear;
}
laplacianSchemes { default Gauss linear corrected; }
interpolationSchemes { default linear; }
snGradSchemes { default corrected; }
fluxRequired { default no; p_rgh; pcorr; alpha1; }
""")
    ps = p_solver or PCG
    _w(os.path.join(case, "system", "fvSolution"), "dictionary", "fvSolution", """
solvers
{
    pcorr { solver PCG; preconditioner DIC; tolerance 1e-10; relTol 0; }
    p_rgh { %s }
    p_rghFinal { %s }
    U { solver PBiCG; preconditioner DILU; tolerance 1e-06; relTol 0; }
}
PIMPLE
{
    momentumPredictor no;
    nCorrectors     3;
    nNonOrthogonalCorrectors 0;
    nAlphaCorr      1;
    nAlphaSubCycles 2;
    cAlpha          1;
}
""" % (ps, ps.replace("relTol 0.05", "relTol 0")))
    _w(os.path.join(case, "constant", "transportProperties"), "dictionary", "transportProperties", """
phase1 { transportModel Newtonian; nu nu [ 0 2 -1 0 0 0 0 ] 1e-06; rho rho [ 1 -3 0 0 0 0 0 ] 1000; }
phase2 { transportModel Newtonian; nu nu [ 0 2 -1 0 0 0 0 ] 1.48e-05; rho rho [ 1 -3 0 0 0 0 0 ] 1; }
sigma sigma [ 1 0 -2 0 0 0 0 ] 0.07;
""")
    _w(os.path.join(case, "constant", "turbulenceProperties"), "dictionary", "turbulenceProperties", "\nsimulationType laminar;\n")
    _w(os.path.join(case, "constant", "g"), "uniformDimensionedVectorField", "g",
       "\ndimensions [0 1 -2 0 0 0 0];\nvalue ( 0 -9.81 0 );\n")
    walls = ("leftWall", "rightWall", "lowerWall")

    def field(name, cls, dims, internal, wall, atm):
        body = "\ndimensions %s;\ninternalField %s;\nboundaryField\n{\n" % (dims, internal)
        for w in walls:
            body += "    %s { %s }\n" % (w, wall)
        body += "    atmosphere { %s }\n    defaultFaces { type empty; }\n}\n" % atm
        _w(os.path.join(case, "0", name), cls, name, body)
    field("U", "volVectorField", "[0 1 -1 0 0 0 0]", "uniform (0 0 0)", "type fixedValue; value uniform (0 0 0);",
          "type pressureInletOutletVelocity; value uniform (0 0 0);")
    field("p_rgh", "volScalarField", "[1 -1 -2 0 0 0 0]", "uniform 0", "type fixedFluxPressure; value uniform 0;",
          "type totalPressure; p0 uniform 0; U U; phi phi; rho rho; psi none; gamma 1; value uniform 0;")
    return field


def write(case, end_time=0.02, libs=None, p_solver=None):
    """the whole case from the committed fixture (mesh + alpha1 of the reference's blockMesh / setFields)"""
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
    field = write_dictionaries(case, end_time, libs, p_solver)
    a = g["alpha1"]
    field("alpha1", "volScalarField", "[0 0 0 0 0 0 0]", "nonuniform List<scalar> %d(%s)" % (a.size, " ".join("%.17g" % v for v in a)),
          "type zeroGradient;", "type inletOutlet; inletValue uniform 0; value uniform 0;")


def run(case, extra_env=None):
    e = env()
    if extra_env:
        e.update(extra_env)
    r = subprocess.run([os.path.join(REF, "interFoam"), "-case", case], env=e, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("interFoam failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    return r.stdout + r.stderr
