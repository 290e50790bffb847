#!/bin/bash
# TEST INFRASTRUCTURE - tier 2 of the reference oracle: the reference's own
# libfiniteVolume units compiled from the sources where they lie under /root/reference (same
# recipe as build_ref.sh), archived STATICALLY so that the linker pulls only the units the driver
# reaches (fvMesh, geometry, interpolation / gradient / laplacian / convection schemes, basic patch
# fields) - the units that need libmeshTools / libtriSurface (mapped and AMI patches, wall distance;
# triSurface needs flex, absent here) are never pulled and nothing stands in for them.
# Used once to pin oracle/fv_oracle.py: tests/golden/make_fv_golden.py runs oracle/_ref/fv_driver and
# commits the vectors.  Not part of __graft_entry__.build() (10 min); outputs only into oracle/_ref/.
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
W="$OUT/build"
JOBS=${JOBS:-8}
if [ ! -d "$REF/src/finiteVolume" ]; then
    echo "build_ref_fv.sh: $REF not present - nothing to do" >&2
    exit 0
fi
[ -f "$OUT/libOpenFOAM.so" ] || bash "$HERE/build_ref.sh"
mkdir -p "$W/fvobj"
for lib in finiteVolume meshTools triSurface fileFormats; do
    d="$W/inc_$lib"
    if [ ! -f "$d/.done" ]; then
        mkdir -p "$d"
        find "$REF/src/$lib" \( -name '*.[CH]' -o -name '*.h' \) -exec ln -sf {} "$d/" \;
        touch "$d/.done"
    fi
done
cpp -P -traditional-cpp -DWM_DP -Dlinux64 "$REF/src/finiteVolume/Make/files" 2>/dev/null | python3 -c '
import sys,re
vars={}
for line in sys.stdin:
    line=line.strip()
    if not line: continue
    m=re.match(r"^(\w+)\s*=\s*(.*)$", line)
    if m:
        v=m.group(2)
        for k,val in vars.items(): v=v.replace("$(%s)"%k,val)
        vars[m.group(1)]=v; continue
    for k,val in vars.items(): line=line.replace("$(%s)"%k,val)
    if line.startswith("LIB") or line.startswith("EXE") or line.endswith(".H"): continue
    print(line)' | sed "s!^!$REF/src/finiteVolume/!" > "$W/fvsources.txt"

CXXFLAGS="-m64 -std=gnu++98 -Dlinux64 -DWM_DP -DNoRepository -ftemplate-depth-100 -O2 -fPIC -w -fpermissive -fno-access-control -I$W/inc_finiteVolume -I$W/inc_meshTools -I$W/inc_triSurface -I$W/inc_fileFormats -I$W/inc"
{
    echo "CXXFLAGS=$CXXFLAGS"
    echo "OBJS="
    i=0
    while read -r src; do
        i=$((i+1))
        o="$W/fvobj/f$i.o"
        echo "OBJS+=$o"
        echo "$o: $src"
        printf '\t@g++ $(CXXFLAGS) -c %s -o %s || echo "FAILED %s" >> %s/fvfailed.txt\n' "$src" "$o" "$src" "$W"
    done < "$W/fvsources.txt"
    echo "all: \$(OBJS)"
} > "$W/Makefile.fv"
rm -f "$W/fvfailed.txt"
if [ -z "$DRIVERS_ONLY" ] || [ ! -f "$W/libfiniteVolume.a" ]; then
make -s -k -C "$W" -f "$W/Makefile.fv" -j"$JOBS" all || true
rm -f "$W/libfiniteVolume.a"
ar rcs "$W/libfiniteVolume.a" "$W"/fvobj/*.o
fi
echo "build_ref_fv.sh: $(ls "$W"/fvobj/*.o | wc -l) units archived; failed: $(cat "$W/fvfailed.txt" 2>/dev/null | wc -l)"
# units whose only entry point is a static registration object (run-time selection of the cyclic patch
# and its patch fields) are not reachable through symbols: name them explicitly
FORCE=""
for u in fvMesh/fvPatches/constraint/cyclic/cyclicFvPatch.C \
         fields/fvPatchFields/constraint/cyclic/cyclicFvPatchFields.C \
         fields/fvPatchFields/basic/fixedValue/fixedValueFvPatchFields.C \
         finiteVolume/gradSchemes/gaussGrad/gaussGrads.C \
         interpolation/surfaceInterpolation/schemes/linear/linear.C \
         fvMatrices/solvers/GAMGSymSolver/GAMGAgglomerations/faceAreaPairGAMGAgglomeration/faceAreaPairGAMGAgglomeration.C \
         fields/fvsPatchFields/constraint/cyclic/cyclicFvsPatchFields.C \
         fvMesh/fvPatches/derived/wall/wallFvPatch.C \
         fvMesh/fvPatches/constraint/empty/emptyFvPatch.C \
         fields/fvPatchFields/constraint/empty/emptyFvPatchFields.C \
         fields/fvsPatchFields/constraint/empty/emptyFvsPatchFields.C; do
    i=$(grep -n "/$u\$" "$W/fvsources.txt" | head -1 | cut -d: -f1)
    [ -n "$i" ] && FORCE="$FORCE $W/fvobj/f$i.o"
done
# the reference's blockMesh library (18 units, src/mesh/blockMesh/Make/files): SURVEY.md 8c tier 2, what turns
# tutorials/incompressible/simpleFoam/pitzDaily/constant/polyMesh/blockMeshDict into config C2's mesh
if [ -f "$HERE/blockmesh_driver.C" ]; then
    mkdir -p "$W/bmobj" "$W/inc_blockMesh"
    if [ ! -f "$W/inc_blockMesh/.done" ]; then
        find "$REF/src/mesh/blockMesh" \( -name '*.[CH]' -o -name '*.h' \) -exec ln -sf {} "$W/inc_blockMesh/" \;
        touch "$W/inc_blockMesh/.done"
    fi
    BMFLAGS="$CXXFLAGS -I$W/inc_blockMesh"
    i=0
    for u in $(grep '\.C$' "$REF/src/mesh/blockMesh/Make/files"); do
        i=$((i+1))
        [ -f "$W/bmobj/b$i.o" ] || g++ $BMFLAGS -c "$REF/src/mesh/blockMesh/$u" -o "$W/bmobj/b$i.o" &
        [ $((i % JOBS)) -eq 0 ] && wait
    done
    # arcEdge holds a cylindricalCS: the coordinate-system units of libmeshTools (the reference's own sources,
    # src/meshTools/coordinateSystems; nothing else of meshTools is reached)
    for u in coordinateSystem.C coordinateSystemNew.C coordinateSystems.C cylindricalCS.C sphericalCS.C toroidalCS.C \
             parabolicCylindricalCS.C coordinateRotation/coordinateRotation.C coordinateRotation/EulerCoordinateRotation.C \
             coordinateRotation/STARCDCoordinateRotation.C; do
        i=$((i+1))
        [ -f "$W/bmobj/b$i.o" ] || g++ $BMFLAGS -c "$REF/src/meshTools/coordinateSystems/$u" -o "$W/bmobj/b$i.o" &
        [ $((i % JOBS)) -eq 0 ] && wait
    done
    wait
    g++ $BMFLAGS -o "$OUT/blockmesh_driver" "$HERE/blockmesh_driver.C" "$W"/bmobj/*.o -L"$OUT" -lOpenFOAM -ldl -lm \
        -Wl,-rpath,'$ORIGIN'
    echo "build_ref_fv.sh: OK -> $OUT/blockmesh_driver ($i units: blockMesh + meshTools/coordinateSystems)"
fi
if [ -f "$HERE/fv_driver.C" ]; then
    g++ $CXXFLAGS -o "$OUT/fv_driver" "$HERE/fv_driver.C" $FORCE "$W/libfiniteVolume.a" -L"$OUT" -lOpenFOAM -ldl -lm \
        -Wl,-rpath,'$ORIGIN'
    echo "build_ref_fv.sh: OK -> $OUT/fv_driver"
fi
# The reference's own icoFoam (BASELINE config C1): applications/solvers/incompressible/icoFoam/icoFoam.C
# compiled where it lies and linked against the archive above + libOpenFOAM.so - an UNCHANGED reference application.
# Units that are only reachable through their run-time selection tables (the schemes and patch fields the cavity's
# dictionaries name) are listed explicitly; nothing stands in for anything.
ICO="$REF/applications/solvers/incompressible/icoFoam"
if [ -f "$ICO/icoFoam.C" ]; then
    FORCE_ICO=""
    for u in fvMesh/fvPatches/derived/wall/wallFvPatch.C \
             fvMesh/fvPatches/constraint/empty/emptyFvPatch.C \
             fields/fvPatchFields/constraint/empty/emptyFvPatchFields.C \
             fields/fvsPatchFields/constraint/empty/emptyFvsPatchFields.C \
             fields/fvPatchFields/basic/fixedValue/fixedValueFvPatchFields.C \
             fields/fvPatchFields/basic/zeroGradient/zeroGradientFvPatchFields.C \
             fields/fvPatchFields/basic/calculated/calculatedFvPatchFields.C \
             fields/fvsPatchFields/basic/calculated/calculatedFvsPatchFields.C \
             finiteVolume/gradSchemes/gaussGrad/gaussGrads.C \
             interpolation/surfaceInterpolation/schemes/linear/linear.C \
             finiteVolume/ddtSchemes/EulerDdtScheme/EulerDdtSchemes.C \
             finiteVolume/convectionSchemes/gaussConvectionScheme/gaussConvectionSchemes.C \
             finiteVolume/laplacianSchemes/gaussLaplacianScheme/gaussLaplacianSchemes.C \
             finiteVolume/snGradSchemes/orthogonalSnGrad/orthogonalSnGrads.C \
             finiteVolume/snGradSchemes/correctedSnGrad/correctedSnGrads.C \
             finiteVolume/divSchemes/gaussDivScheme/gaussDivSchemes.C \
             fvMatrices/solvers/GAMGSymSolver/GAMGAgglomerations/faceAreaPairGAMGAgglomeration/faceAreaPairGAMGAgglomeration.C; do
        i=$(grep -n "/$u\$" "$W/fvsources.txt" | head -1 | cut -d: -f1)
        if [ -n "$i" ] && [ -f "$W/fvobj/f$i.o" ]; then FORCE_ICO="$FORCE_ICO $W/fvobj/f$i.o"
        else echo "build_ref_fv.sh: unit $u is not built - icoFoam cannot be linked (no stand-ins)" >&2; FORCE_ICO="MISSING"; break; fi
    done
    if [ "$FORCE_ICO" != "MISSING" ]; then
        g++ $CXXFLAGS -I"$ICO" -c "$ICO/icoFoam.C" -o "$W/icoFoam.o"
        g++ -o "$OUT/icoFoam" "$W/icoFoam.o" $FORCE_ICO "$W/libfiniteVolume.a" -L"$OUT" -lOpenFOAM -ldl -lm -Wl,-rpath,'$ORIGIN'
        echo "build_ref_fv.sh: OK -> $OUT/icoFoam (the reference's icoFoam.C, unchanged)"
    fi
fi
# The reference's own simpleFoam (BASELINE config C2; SURVEY 8d tier 3): applications/solvers/incompressible/simpleFoam/
# simpleFoam.C compiled where it lies, with the units of libincompressibleRASModels / libincompressibleTurbulenceModel /
# libincompressibleTransportModels / libfvOptions / libmeshTools / libsampling it reaches for the pitzDaily case
# (kEpsilon + its wall functions, Newtonian transport, the fvOption list, cellDistFuncs / cellSet / tetOverlapVolume,
# meshToMeshNew) - 27 units of the reference, compiled from their own sources; nothing stands in for anything.
SIM="$REF/applications/solvers/incompressible/simpleFoam"
if [ -f "$SIM/simpleFoam.C" ] && [ -z "$NO_SIMPLEFOAM" ]; then
    for lib in turbulenceModels/incompressible/RAS turbulenceModels/incompressible/turbulenceModel transportModels/incompressible fvOptions sampling; do
        d="$W/inc_$(echo $lib | tr '/' '_')"
        if [ ! -f "$d/.done" ]; then
            mkdir -p "$d"
            find "$REF/src/$lib" \( -name '*.[CH]' -o -name '*.h' \) -exec ln -sf {} "$d/" \;
            touch "$d/.done"
        fi
    done
    SFLAGS="$CXXFLAGS -I$W/inc_turbulenceModels_incompressible_RAS -I$W/inc_turbulenceModels_incompressible_turbulenceModel -I$W/inc_transportModels_incompressible -I$W/inc_fvOptions -I$W/inc_sampling -I$REF/src/turbulenceModels -I$REF/src/transportModels"
    mkdir -p "$W/sfobj"
    i=0
    for u in turbulenceModels/incompressible/turbulenceModel/turbulenceModel.C \
             turbulenceModels/incompressible/turbulenceModel/laminar/laminar.C \
             turbulenceModels/incompressible/RAS/RASModel/RASModel.C \
             turbulenceModels/incompressible/RAS/laminar/laminar.C \
             turbulenceModels/incompressible/RAS/kEpsilon/kEpsilon.C \
             turbulenceModels/incompressible/RAS/derivedFvPatchFields/wallFunctions/nutWallFunctions/nutWallFunction/nutWallFunctionFvPatchScalarField.C \
             turbulenceModels/incompressible/RAS/derivedFvPatchFields/wallFunctions/nutWallFunctions/nutkWallFunction/nutkWallFunctionFvPatchScalarField.C \
             turbulenceModels/incompressible/RAS/derivedFvPatchFields/wallFunctions/nutWallFunctions/nutLowReWallFunction/nutLowReWallFunctionFvPatchScalarField.C \
             turbulenceModels/incompressible/RAS/derivedFvPatchFields/wallFunctions/epsilonWallFunctions/epsilonWallFunction/epsilonWallFunctionFvPatchScalarField.C \
             turbulenceModels/incompressible/RAS/derivedFvPatchFields/wallFunctions/omegaWallFunctions/omegaWallFunction/omegaWallFunctionFvPatchScalarField.C \
             turbulenceModels/incompressible/RAS/derivedFvPatchFields/wallFunctions/kqRWallFunctions/kqRWallFunction/kqRWallFunctionFvPatchFields.C \
             turbulenceModels/incompressible/RAS/backwardsCompatibility/wallFunctions/backwardsCompatibilityWallFunctions.C \
             transportModels/incompressible/viscosityModels/viscosityModel/viscosityModel.C \
             transportModels/incompressible/viscosityModels/viscosityModel/viscosityModelNew.C \
             transportModels/incompressible/viscosityModels/Newtonian/Newtonian.C \
             transportModels/incompressible/transportModel/transportModel.C \
             transportModels/incompressible/singlePhaseTransportModel/singlePhaseTransportModel.C \
             fvOptions/fvOptions/fvOption.C fvOptions/fvOptions/fvOptionIO.C fvOptions/fvOptions/fvOptionList.C \
             fvOptions/fvOptions/fvIOoptionList.C \
             meshTools/cellDist/cellDistFuncs.C meshTools/sets/topoSets/topoSet.C meshTools/sets/topoSets/cellSet.C \
             meshTools/tetOverlapVolume/tetOverlapVolume.C \
             sampling/meshToMeshInterpolation/meshToMeshNew/meshToMeshNewParallelOps.C \
             sampling/meshToMeshInterpolation/meshToMeshNew/meshToMeshNew.C; do
        i=$((i+1))
        [ -f "$W/sfobj/s$i.o" ] || g++ $SFLAGS -c "$REF/src/$u" -o "$W/sfobj/s$i.o" &
        [ $((i % JOBS)) -eq 0 ] && wait
    done
    wait
    FORCE_SIM=""
    for u in fvMesh/fvPatches/derived/wall/wallFvPatch.C \
             fvMesh/fvPatches/constraint/empty/emptyFvPatch.C \
             fields/fvPatchFields/constraint/empty/emptyFvPatchFields.C \
             fields/fvsPatchFields/constraint/empty/emptyFvsPatchFields.C \
             fields/fvPatchFields/basic/fixedValue/fixedValueFvPatchFields.C \
             fields/fvPatchFields/basic/zeroGradient/zeroGradientFvPatchFields.C \
             fields/fvPatchFields/basic/calculated/calculatedFvPatchFields.C \
             fields/fvsPatchFields/basic/calculated/calculatedFvsPatchFields.C \
             finiteVolume/gradSchemes/gaussGrad/gaussGrads.C \
             interpolation/surfaceInterpolation/schemes/linear/linear.C \
             interpolation/surfaceInterpolation/limitedSchemes/upwind/upwind.C \
             finiteVolume/ddtSchemes/steadyStateDdtScheme/steadyStateDdtSchemes.C \
             finiteVolume/convectionSchemes/gaussConvectionScheme/gaussConvectionSchemes.C \
             finiteVolume/convectionSchemes/boundedConvectionScheme/boundedConvectionSchemes.C \
             finiteVolume/laplacianSchemes/gaussLaplacianScheme/gaussLaplacianSchemes.C \
             finiteVolume/snGradSchemes/correctedSnGrad/correctedSnGrads.C \
             finiteVolume/divSchemes/gaussDivScheme/gaussDivSchemes.C \
             fvMatrices/solvers/GAMGSymSolver/GAMGAgglomerations/faceAreaPairGAMGAgglomeration/faceAreaPairGAMGAgglomeration.C; do
        i=$(grep -n "/$u\$" "$W/fvsources.txt" | head -1 | cut -d: -f1)
        if [ -n "$i" ] && [ -f "$W/fvobj/f$i.o" ]; then FORCE_SIM="$FORCE_SIM $W/fvobj/f$i.o"
        else echo "build_ref_fv.sh: unit $u is not built - simpleFoam cannot be linked (no stand-ins)" >&2; FORCE_SIM="MISSING"; break; fi
    done
    if [ "$FORCE_SIM" != "MISSING" ]; then
        g++ $SFLAGS -I"$SIM" -c "$SIM/simpleFoam.C" -o "$W/simpleFoam.o"
        # (-rdynamic: the executable exports its symbols - the run-time selection tables of the statically linked finiteVolume
        #  units among them - so that a library loaded through `libs (...)` registers into THOSE tables, as it would with
        #  the reference's shared libfiniteVolume.so; a link flag, simpleFoam.C is compiled unchanged)
        g++ -rdynamic -o "$OUT/simpleFoam" "$W/simpleFoam.o" "$W"/sfobj/*.o $FORCE_SIM "$W/libfiniteVolume.a" -L"$OUT" -lOpenFOAM -ldl -lm -Wl,-rpath,'$ORIGIN' \
            && echo "build_ref_fv.sh: OK -> $OUT/simpleFoam (the reference's simpleFoam.C, unchanged)"
    fi
fi
