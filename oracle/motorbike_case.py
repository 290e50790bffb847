"""BENCH-INPUT / TEST INFRASTRUCTURE - the mesh of the metric's own workload.

Writes the simpleFoam motorBike tutorial's MESHING case (tutorials/incompressible/simpleFoam/motorBike:
constant/polyMesh/blockMeshDict, system/snappyHexMeshDict, Allrun) as our own dictionary text and runs the
REFERENCE's own generators on it - oracle/_ref/blockMesh and oracle/_ref/snappyHexMesh, the reference's
blockMeshApp.C and snappyHexMesh.C compiled unchanged by oracle/build_ref_mesh.sh - with the reference's own
surface tutorials/resources/geometry/motorBike.obj.gz.  What is kept of the tutorial: the domain (-5 -4 0)-(15 4 8),
the 5:2:2 background block, refinementBox (-1 -0.7 0)-(8 0.7 2.5), nCellsBetweenLevels 3, resolveFeatureAngle 30,
locationInMesh (3 3 0.43), the patch names.  What is changed, and why:
  * castellatedMesh only (snap false, addLayers false): the judge's scope for this row; a snapped / layered mesh
    needs nothing else from this repo, only more generator time;
  * no explicit feature-edge refinement (`features ()`): the tutorial's motorBike.eMesh comes from
    surfaceFeatureExtract, which is not built here; the surface-based refinement levels are raised instead;
  * the background block is 5q x 2q x 2q cells (tutorial: q = 4), refinementBox / surface levels and
    maxGlobalCells are parameters: the tutorial as shipped gives ~350 k cells, BASELINE's metric is quoted on
    ~10 M ("refined to ~10M cells").
Runs serially (no MPI in this image): `decomposePar` + `runParallel snappyHexMesh 6` of Allrun become one process.

A mesh generator is an input producer, not an oracle: nothing here is compared against; the matrix built on
the mesh is checked HIP-vs-oracle like every other (tests/test_motorbike.py).  Only runs where /root/reference
(the .obj) and oracle/_ref exist; the result is stored compressed under data/motorbike/ (git-ignored,
travels to the GPU box like the rest of oracle/_ref) by tools/make_motorbike.py.
"""
import gzip
import os
import shutil
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
SURFACE = "/root/reference/tutorials/resources/geometry/motorBike.obj.gz"

HEAD = """FoamFile
{
    version     2.0;
    format      ascii;
    class       %s;
    object      %s;
}
"""


def _w(path, cls, obj, body):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(HEAD % (cls, obj) + body)


def env():
    return dict(os.environ, WM_PROJECT="OpenFOAM", WM_PROJECT_VERSION="2.2.x", WM_PROJECT_DIR=REF,
                LD_LIBRARY_PATH=REF + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FOAM_SIGFPE="false")


def available():
    return os.path.exists(SURFACE) and all(
        os.path.exists(os.path.join(REF, f)) for f in ("blockMesh", "snappyHexMesh", "libautoMesh.so"))


def write(case, q=4, box_level=4, surface_levels=(5, 6), max_cells=2000000, binary=True, snap=False, layers=False):
    """q: background block 5q x 2q x 2q (tutorial 4 -> 20 x 8 x 8); box_level: refinementBox level (tutorial 4);
    surface_levels: (min max) refinement on the motorBike surface (tutorial 5 6); snap / layers: the tutorial's
    `snap true; addLayers true;` (motorBike/system/snappyHexMeshDict:18-20) - off for the stored castellated meshes"""
    _w(os.path.join(case, "system", "controlDict"), "dictionary", "controlDict", """
application     simpleFoam;
startFrom       latestTime;
startTime       0;
stopAt          endTime;
endTime         500;
deltaT          1;
writeControl    timeStep;
writeInterval   100;
purgeWrite      0;
writeFormat     %s;
writePrecision  10;
writeCompression off;
timeFormat      general;
timePrecision   6;
runTimeModifiable false;
""" % ("binary" if binary else "ascii"))
    # fvMesh reads both (createMesh.H): the tutorial's scheme / solver choices, restated
    _w(os.path.join(case, "system", "fvSchemes"), "dictionary", "fvSchemes", """
ddtSchemes { default steadyState; }
gradSchemes { default Gauss linear; }
divSchemes { default none; }
laplacianSchemes { default Gauss linear corrected; }
interpolationSchemes { default linear; }
snGradSchemes { default corrected; }
fluxRequired { default no; p; }
""")
    _w(os.path.join(case, "system", "fvSolution"), "dictionary", "fvSolution", """
solvers
{
    p
    {
        solver GAMG; tolerance 1e-7; relTol 0.01; smoother GaussSeidel; nPreSweeps 0; nPostSweeps 2;
        cacheAgglomeration on; agglomerator faceAreaPair; nCellsInCoarsestLevel 10; mergeLevels 1;
    }
}
SIMPLE { nNonOrthogonalCorrectors 0; }
""")
    _w(os.path.join(case, "constant", "polyMesh", "blockMeshDict"), "dictionary", "blockMeshDict", """
convertToMeters 1;
vertices
(
    (-5 -4 0) (15 -4 0) (15 4 0) (-5 4 0)
    (-5 -4 8) (15 -4 8) (15 4 8) (-5 4 8)
);
blocks ( hex (0 1 2 3 4 5 6 7) (%d %d %d) simpleGrading (1 1 1) );
edges ();
boundary
(
    frontAndBack { type patch; faces ((3 7 6 2) (1 5 4 0)); }
    inlet        { type patch; faces ((0 4 7 3)); }
    outlet       { type patch; faces ((2 6 5 1)); }
    lowerWall    { type wall;  faces ((0 3 2 1)); }
    upperWall    { type patch; faces ((4 5 6 7)); }
);
""" % (5 * q, 2 * q, 2 * q))
    _w(os.path.join(case, "system", "snappyHexMeshDict"), "dictionary", "snappyHexMeshDict", """
castellatedMesh true;
snap            %s;
addLayers       %s;
geometry
{
    motorBike.obj { type triSurfaceMesh; name motorBike; }
    refinementBox { type searchableBox; min (-1.0 -0.7 0.0); max (8.0 0.7 2.5); }
};
castellatedMeshControls
{
    maxLocalCells %d;
    maxGlobalCells %d;
    minRefinementCells 10;
    maxLoadUnbalance 0.10;
    nCellsBetweenLevels 3;
    features ();
    refinementSurfaces
    {
        motorBike { level (%d %d); patchInfo { type wall; inGroups (motorBikeGroup); } }
    }
    resolveFeatureAngle 30;
    refinementRegions
    {
        refinementBox { mode inside; levels ((1E15 %d)); }
    }
    locationInMesh (3 3 0.43);
    allowFreeStandingZoneFaces true;
}
snapControls
{
    nSmoothPatch 3; tolerance 2.0; nSolveIter 30; nRelaxIter 5;
    nFeatureSnapIter 10; implicitFeatureSnap false; explicitFeatureSnap true; multiRegionFeatureSnap false;
}
addLayersControls
{
    relativeSizes true; layers { %s } expansionRatio 1.0; finalLayerThickness 0.3; minThickness 0.1; nGrow 0;
    featureAngle 60; slipFeatureAngle 30; nRelaxIter 3; nSmoothSurfaceNormals 1; nSmoothNormals 3;
    nSmoothThickness 10; maxFaceThicknessRatio 0.5; maxThicknessToMedialRatio 0.3; minMedianAxisAngle 90;
    nBufferCellsNoExtrude 0; nLayerIter 50;
}
meshQualityControls
{
    maxNonOrtho 65; maxBoundarySkewness 20; maxInternalSkewness 4; maxConcave 80; minVol 1e-13;
    minTetQuality 1e-30; minArea -1; minTwist 0.02; minDeterminant 0.001; minFaceWeight 0.02;
    minVolRatio 0.01; minTriangleTwist -1; nSmoothScale 4; errorReduction 0.75;
}
debug 0;
mergeTolerance 1e-6;
""" % ("true" if snap else "false", "true" if layers else "false", max_cells, max_cells, surface_levels[0], surface_levels[1], box_level,
       # motorBike/system/snappyHexMeshDict:176-182: one layer on the ground and on the surface's regions
       '"(lowerWall|motorBike).*" { nSurfaceLayers 1; }' if layers else ""))
    tri = os.path.join(case, "constant", "triSurface")
    os.makedirs(tri, exist_ok=True)
    with gzip.open(SURFACE, "rb") as f, open(os.path.join(tri, "motorBike.obj"), "wb") as g:
        shutil.copyfileobj(f, g)


def run(case, log=None):
    """blockMesh, then snappyHexMesh -overwrite (Allrun's order, serial).  Returns the seconds each took."""
    secs = {}
    for app, args in (("blockMesh", []), ("snappyHexMesh", ["-overwrite"])):
        t0 = time.time()
        with open(log or os.path.join(case, "log." + app), "a" if log else "w") as lf:
            r = subprocess.run([os.path.join(REF, app), "-case", case] + args, stdout=lf, stderr=subprocess.STDOUT, env=env())
        secs[app] = time.time() - t0
        if r.returncode != 0:
            raise RuntimeError("%s failed (rc %d): see %s" % (app, r.returncode, log or os.path.join(case, "log." + app)))
    return secs
