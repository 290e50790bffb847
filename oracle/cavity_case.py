"""TEST INFRASTRUCTURE - writes BASELINE config C1, the icoFoam lid-driven cavity, as an OpenFOAM case directory.

The numbers are those of the reference's tutorial (tutorials/incompressible/icoFoam/cavity: system/controlDict,
system/fvSchemes, system/fvSolution, constant/transportProperties, constant/polyMesh/blockMeshDict, 0/U, 0/p), restated
here as our own dictionary text so that the case can be created where /root/reference does not exist (the GPU box):
unit square x 0.1 (convertToMeters 0.1), n x n x 1 cells, lid (1 0 0), nu 0.01, Euler / Gauss linear / orthogonal,
p: PCG + DIC 1e-6, U: PBiCG + DILU 1e-5, PISO 2 correctors, pRefCell 0.
The mesh itself comes from the reference's own blockMesh library (oracle/_ref/blockmesh_driver)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")

HEAD = """FoamFile
{
    version     2.0;
    format      ascii;
    class       %s;
    %sobject      %s;
}
"""


def _w(path, cls, obj, body, location=None):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    loc = ('location    "%s";\n    ' % location) if location else ""
    with open(path, "w") as f:
        f.write(HEAD % (cls, loc, obj) + body)


def env():
    return dict(os.environ, WM_PROJECT="OpenFOAM", WM_PROJECT_VERSION="2.2.x", WM_PROJECT_DIR=REF,
                LD_LIBRARY_PATH=REF + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FOAM_SIGFPE="false")


def available():
    return all(os.path.exists(os.path.join(REF, f)) for f in ("icoFoam", "blockmesh_driver", "libOpenFOAM.so"))


def write(case, n=40, steps=100, libs=None, p_solver=None):
    """n x n cavity, `steps` time steps at Courant-preserving deltaT = 0.005 * 20 / n; libs: list of plugin libraries
    for `libs (...)` in controlDict; p_solver: replacement text of the p solver block (default: the tutorial's PCG/DIC)"""
    dt = 0.005 * 20.0 / n
    _w(os.path.join(case, "system", "controlDict"), "dictionary", "controlDict", """
application     icoFoam;
startFrom       startTime;
startTime       0;
stopAt          endTime;
endTime         %.10g;
deltaT          %.10g;
writeControl    timeStep;
writeInterval   100000;
purgeWrite      0;
writeFormat     ascii;
writePrecision  6;
writeCompression off;
timeFormat      general;
timePrecision   6;
runTimeModifiable false;
%s
""" % (dt * steps, dt, ("libs (%s);" % " ".join('"%s"' % l for l in libs)) if libs else ""), "system")
    _w(os.path.join(case, "system", "fvSchemes"), "dictionary", "fvSchemes", """
ddtSchemes { default Euler; }
gradSchemes { default Gauss linear; grad(p) Gauss linear; }
divSchemes { default none; div(phi,U) Gauss linear; }
laplacianSchemes { default none; laplacian(nu,U) Gauss linear orthogonal; laplacian((1|A(U)),p) Gauss linear orthogonal; }
interpolationSchemes { default linear; interpolate(HbyA) linear; }
snGradSchemes { default orthogonal; }
fluxRequired { default no; p ; }
""", "system")
    _w(os.path.join(case, "system", "fvSolution"), "dictionary", "fvSolution", """
solvers
{
    p
    {
%s
    }
    U
    {
        solver          PBiCG;
        preconditioner  DILU;
        tolerance       1e-05;
        relTol          0;
    }
}
PISO
{
    nCorrectors     2;
    nNonOrthogonalCorrectors 0;
    pRefCell        0;
    pRefValue       0;
}
""" % (p_solver or "        solver          PCG;\n        preconditioner  DIC;\n        tolerance       1e-06;\n        relTol          0;"),
       "system")
    _w(os.path.join(case, "constant", "transportProperties"), "dictionary", "transportProperties",
       "\nnu              nu [ 0 2 -1 0 0 0 0 ] 0.01;\n", "constant")
    _w(os.path.join(case, "constant", "polyMesh", "blockMeshDict"), "dictionary", "blockMeshDict", """
convertToMeters 0.1;
vertices ( (0 0 0) (1 0 0) (1 1 0) (0 1 0) (0 0 0.1) (1 0 0.1) (1 1 0.1) (0 1 0.1) );
blocks ( hex (0 1 2 3 4 5 6 7) (%d %d 1) simpleGrading (1 1 1) );
edges ( );
boundary
(
    movingWall { type wall; faces ( (3 7 6 2) ); }
    fixedWalls { type wall; faces ( (0 4 7 3) (2 6 5 1) (1 5 4 0) ); }
    frontAndBack { type empty; faces ( (0 3 2 1) (4 5 6 7) ); }
);
mergePatchPairs ( );
""" % (n, n))
    _w(os.path.join(case, "0", "U"), "volVectorField", "U", """
dimensions      [0 1 -1 0 0 0 0];
internalField   uniform (0 0 0);
boundaryField
{
    movingWall { type fixedValue; value uniform (1 0 0); }
    fixedWalls { type fixedValue; value uniform (0 0 0); }
    frontAndBack { type empty; }
}
""")
    _w(os.path.join(case, "0", "p"), "volScalarField", "p", """
dimensions      [0 2 -2 0 0 0 0];
internalField   uniform 0;
boundaryField
{
    movingWall { type zeroGradient; }
    fixedWalls { type zeroGradient; }
    frontAndBack { type empty; }
}
""")
    r = subprocess.run([os.path.join(REF, "blockmesh_driver"), case], env=env(), capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("blockmesh_driver failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


def run(case, extra_env=None):
    """the reference's own icoFoam (oracle/_ref/icoFoam, linked from applications/solvers/incompressible/icoFoam/icoFoam.C by
    oracle/build_ref_fv.sh) on the case; returns its log"""
    e = env()
    if extra_env:
        e.update(extra_env)
    r = subprocess.run([os.path.join(REF, "icoFoam"), "-case", case], env=e, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("icoFoam failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    return r.stdout + r.stderr


def solve_lines(log):
    """[(solverName, field, initialResidual, finalResidual, nIterations)] of every `Solving for` line, in order"""
    out = []
    for l in log.splitlines():
        if ":  Solving for " in l:
            name, rest = l.split(":  Solving for ")
            f = rest.replace(",", " ").split()
            out.append((name.strip(), f[0], float(f[4]), float(f[8]), int(f[11])))
    return out
