#!/bin/bash
# Builds libhipLduSolvers.so against the reference headers (flat include dir made by
# oracle/build_ref.sh) and links it with libldugpu.so.  Needs /root/reference at build time
# only; the binary travels to the GPU box next to oracle/_ref/libOpenFOAM.so.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
INC="$ROOT/oracle/_ref/build/inc"
if [ ! -d "$INC" ]; then
    echo "build_plugin.sh: reference headers not available ($INC) - keeping prebuilt plugin" >&2
    exit 0
fi
mkdir -p "$HERE/../lib"
g++ -m64 -std=gnu++98 -Dlinux64 -DWM_DP -DNoRepository -ftemplate-depth-100 -O2 -fPIC -w -fpermissive \
    -fno-access-control -I"$INC" -I"$ROOT/include" -shared -o "$HERE/../lib/libhipLduSolvers.so" \
    "$HERE/hipLduSolvers.C" -L"$HERE/../lib" -lldugpu -L"$ROOT/oracle/_ref" -lOpenFOAM \
    -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$ROOT/oracle/_ref"
echo "build_plugin.sh: OK -> $HERE/../lib/libhipLduSolvers.so"
# the scheme plug-in (hipGauss laplacian / convection / gradient schemes): needs the libfiniteVolume headers collected by
# oracle/build_ref_fv.sh; its finiteVolume symbols are resolved at load time by the application that loads it
FVINC="$ROOT/oracle/_ref/build/inc_finiteVolume"
if [ -d "$FVINC" ]; then
    if [ ! -f "$HERE/../lib/libhipFvSchemes.so" ] || [ "$HERE/hipFvSchemes.C" -nt "$HERE/../lib/libhipFvSchemes.so" ] \
       || [ "$ROOT/include/ldugpu.h" -nt "$HERE/../lib/libhipFvSchemes.so" ]; then
        g++ -m64 -std=gnu++98 -Dlinux64 -DWM_DP -DNoRepository -ftemplate-depth-100 -O2 -fPIC -w -fpermissive \
            -fno-access-control -I"$FVINC" -I"$ROOT/oracle/_ref/build/inc_meshTools" -I"$INC" -I"$ROOT/include" -shared \
            -o "$HERE/../lib/libhipFvSchemes.so" "$HERE/hipFvSchemes.C" -L"$HERE/../lib" -lhipLduSolvers -lldugpu \
            -L"$ROOT/oracle/_ref" -lOpenFOAM -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$ROOT/oracle/_ref"
    fi
    echo "build_plugin.sh: OK -> $HERE/../lib/libhipFvSchemes.so"
else
    echo "build_plugin.sh: libfiniteVolume headers not available ($FVINC) - keeping prebuilt libhipFvSchemes.so" >&2
fi
