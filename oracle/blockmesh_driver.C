// TEST INFRASTRUCTURE - runs the REFERENCE's own blockMesh library (src/mesh/blockMesh, 18 units compiled from
// /root/reference by oracle/build_ref_fv.sh) on a case's constant/polyMesh/blockMeshDict and writes the polyMesh the
// way the blockMesh application does (applications/utilities/mesh/generation/blockMesh/blockMeshApp.C:150-330:
// blockMesh(dict) -> polyMesh(points, cellShapes, patches, names, dicts, defaultFaces) -> write with 10 digits).
// mergePatchPairs / cell zones are not handled (the application needs libdynamicMesh / libmeshTools for them).
// Our code; only reference HEADERS are included.  Never shipped.
#include "Time.H"
#include "IOdictionary.H"
#include "blockMesh.H"
#include "polyMesh.H"
#include "emptyPolyPatch.H"
#include "OSspecific.H"
#include <cstdio>

using namespace Foam;

int main(int argc, char* argv[])
{
    if (argc != 2) { fprintf(stderr, "usage: blockmesh_driver <caseDir>\n"); return 2; }
    fileName caseDir(argv[1]);
    Time runTime(Time::controlDictName, fileName(caseDir.path()), fileName(caseDir.name()));
    const word regionName(polyMesh::defaultRegion);
    IOdictionary meshDict
    (
        IOobject("blockMeshDict", runTime.constant(), polyMesh::meshSubDir, runTime, IOobject::MUST_READ, IOobject::NO_WRITE, false)
    );
    if (meshDict.found("mergePatchPairs") && List<Pair<word> >(meshDict.lookup("mergePatchPairs")).size())
    {
        FatalErrorIn("blockmesh_driver") << "mergePatchPairs are not supported by this driver" << exit(FatalError);
    }
    blockMesh blocks(meshDict, regionName);
    word defaultFacesName = "defaultFaces";
    word defaultFacesType = emptyPolyPatch::typeName;
    polyMesh mesh
    (
        IOobject(regionName, runTime.constant(), runTime),
        xferCopy(blocks.points()),
        blocks.cells(),
        blocks.patches(),
        blocks.patchNames(),
        blocks.patchDicts(),
        defaultFacesName,
        defaultFacesType
    );
    IOstream::defaultPrecision(max(10u, IOstream::defaultPrecision()));
    mesh.removeFiles();
    if (!mesh.write())
    {
        FatalErrorIn("blockmesh_driver") << "Failed writing polyMesh." << exit(FatalError);
    }
    Info<< "blockmesh_driver: " << mesh.nCells() << " cells, " << mesh.nInternalFaces() << " internal faces" << endl;
    return 0;
}
