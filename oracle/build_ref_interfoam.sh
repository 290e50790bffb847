#!/bin/bash
# TEST INFRASTRUCTURE - the reference's own interFoam (BASELINE config C5) and setFields, compiled
# where they lie under /root/reference with the recipe of build_ref.sh / build_ref_fv.sh / build_ref_mesh.sh:
#   applications/solvers/multiphase/interFoam/interFoam.C               -> oracle/_ref/interFoam   (UNCHANGED)
#   applications/utilities/preProcessing/setFields/setFields.C          -> oracle/_ref/setFields   (UNCHANGED)
# with the units of libtwoPhaseMixture, libinterfaceProperties, libtwoPhaseProperties, libincompressibleTransportModels,
# libincompressibleTurbulenceModel and the four fvOption list units + the two meshToMeshNew units of libsampling they
# reach, linked against the shared libraries build_ref_mesh.sh makes (libfiniteVolume.so, libmeshTools.so, ...).
# Nothing stands in for anything.  Outputs only into oracle/_ref/.
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
W="$OUT/build"
JOBS=${JOBS:-8}
if [ ! -d "$REF/applications/solvers/multiphase/interFoam" ]; then
    echo "build_ref_interfoam.sh: $REF not present - nothing to do" >&2
    exit 0
fi
[ -f "$OUT/libmeshTools.so" ] || bash "$HERE/build_ref_mesh.sh"
INC="-I$W/inc -I$W/inc_finiteVolume -I$W/inc_meshTools -I$W/inc_triSurface -I$W/inc_fileFormats -I$W/inc_surfMesh"
for lib in transportModels/twoPhaseMixture transportModels/interfaceProperties transportModels/twoPhaseProperties \
           transportModels/incompressible turbulenceModels/incompressible/turbulenceModel fvOptions sampling; do
    d="$W/inc_$(echo $lib | tr '/' '_')"
    if [ ! -f "$d/.done" ]; then
        mkdir -p "$d"
        find "$REF/src/$lib" \( -name '*.[CH]' -o -name '*.h' \) -exec ln -sf {} "$d/" \;
        touch "$d/.done"
    fi
    INC="$INC -I$d"
done
CXXFLAGS="-m64 -std=gnu++98 -Dlinux64 -DWM_DP -DNoRepository -ftemplate-depth-100 -O2 -fPIC -w -fpermissive -fno-access-control $INC -I$REF/src/transportModels -I$REF/src/turbulenceModels"
mkdir -p "$W/ifobj"
UNITS="transportModels/twoPhaseMixture/twoPhaseMixture/twoPhaseMixture.C
transportModels/interfaceProperties/interfaceProperties.C
transportModels/interfaceProperties/interfaceCompression/interfaceCompression.C
transportModels/twoPhaseProperties/alphaContactAngle/alphaContactAngle/alphaContactAngleFvPatchScalarField.C
transportModels/twoPhaseProperties/alphaContactAngle/constantAlphaContactAngle/constantAlphaContactAngleFvPatchScalarField.C
transportModels/twoPhaseProperties/alphaContactAngle/dynamicAlphaContactAngle/dynamicAlphaContactAngleFvPatchScalarField.C
transportModels/twoPhaseProperties/alphaContactAngle/timeVaryingAlphaContactAngle/timeVaryingAlphaContactAngleFvPatchScalarField.C
transportModels/twoPhaseProperties/alphaFixedPressure/alphaFixedPressureFvPatchScalarField.C
transportModels/incompressible/viscosityModels/viscosityModel/viscosityModel.C
transportModels/incompressible/viscosityModels/viscosityModel/viscosityModelNew.C
transportModels/incompressible/viscosityModels/Newtonian/Newtonian.C
transportModels/incompressible/viscosityModels/powerLaw/powerLaw.C
transportModels/incompressible/viscosityModels/CrossPowerLaw/CrossPowerLaw.C
transportModels/incompressible/viscosityModels/BirdCarreau/BirdCarreau.C
transportModels/incompressible/viscosityModels/HerschelBulkley/HerschelBulkley.C
transportModels/incompressible/transportModel/transportModel.C
transportModels/incompressible/singlePhaseTransportModel/singlePhaseTransportModel.C
transportModels/incompressible/incompressibleTwoPhaseMixture/incompressibleTwoPhaseMixture.C
turbulenceModels/incompressible/turbulenceModel/turbulenceModel.C
turbulenceModels/incompressible/turbulenceModel/laminar/laminar.C
fvOptions/fvOptions/fvOption.C
fvOptions/fvOptions/fvOptionIO.C
fvOptions/fvOptions/fvOptionList.C
fvOptions/fvOptions/fvIOoptionList.C
sampling/meshToMeshInterpolation/meshToMeshNew/meshToMeshNewParallelOps.C
sampling/meshToMeshInterpolation/meshToMeshNew/meshToMeshNew.C"
i=0
for u in $UNITS; do
    i=$((i+1))
    [ -f "$W/ifobj/i$i.o" ] || g++ $CXXFLAGS -c "$REF/src/$u" -o "$W/ifobj/i$i.o" &
    [ $((i % JOBS)) -eq 0 ] && wait
done
wait
LINK="-Wl,--no-as-needed -L$OUT -lmeshTools -lsurfMesh -ltriSurface -lfileFormats -lfiniteVolume -lOpenFOAM -ldl -lm -Wl,-rpath,\$ORIGIN -Wl,--allow-shlib-undefined"
IF="$REF/applications/solvers/multiphase/interFoam"
g++ $CXXFLAGS -I"$IF" -c "$IF/interFoam.C" -o "$W/interFoam.o"
g++ -o "$OUT/interFoam" "$W/interFoam.o" "$W"/ifobj/*.o $LINK && echo "build_ref_interfoam.sh: OK -> $OUT/interFoam (the reference's interFoam.C, unchanged)"
SF="$REF/applications/utilities/preProcessing/setFields"
g++ $CXXFLAGS -I"$SF" -c "$SF/setFields.C" -o "$W/setFields.o"
g++ -o "$OUT/setFields" "$W/setFields.o" $LINK && echo "build_ref_interfoam.sh: OK -> $OUT/setFields (the reference's setFields.C, unchanged)"
