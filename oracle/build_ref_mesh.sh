#!/bin/bash
# TEST / BENCH-INPUT INFRASTRUCTURE - the reference's own mesh generators, built from the sources where
# they lie under /root/reference (same recipe as build_ref.sh / build_ref_fv.sh: g++ straight on the
# units of each library's Make/files, no wmake):
#   applications/utilities/mesh/generation/blockMesh/blockMeshApp.C        -> oracle/_ref/blockMesh
#   applications/utilities/mesh/generation/snappyHexMesh/snappyHexMesh.C   -> oracle/_ref/snappyHexMesh
# and the shared libraries they link, as the reference builds them (one .so per library):
#   fileFormats triSurface surfMesh meshTools edgeMesh extrudeModel dynamicMesh lagrangian
#   decompositionMethods distributed autoMesh blockMesh finiteVolume
# The two flex units (triSurface/.../readSTLASCII.L, surfMesh/.../STLsurfaceFormatASCII.L) cannot be
# generated here (no flex) and are simply NOT built: the libraries keep the two functions as undefined
# (lazily bound) symbols, exactly like a shared library whose dependency is missing; nothing stands in
# for them, and reading an ASCII .stl would abort in the dynamic linker.  The motorBike tutorial's
# surface is a Wavefront .obj (triSurface/interfaces/OBJ/readOBJ.C), which does not reach them.
# A mesh generator is an input producer for bench.py --mesh motorbike, not an oracle.
# Outputs only into oracle/_ref/ (git-ignored).  ~500 units, 20-40 min on 8 cores.
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
W="$OUT/build"
JOBS=${JOBS:-8}
if [ ! -d "$REF/src/mesh/autoMesh" ]; then
    echo "build_ref_mesh.sh: $REF not present - nothing to do" >&2
    exit 0
fi
[ -f "$OUT/libOpenFOAM.so" ] || bash "$HERE/build_ref.sh"
[ -f "$W/libfiniteVolume.a" ] || DRIVERS_ONLY= bash "$HERE/build_ref_fv.sh"

# name : directory under src/
LIBS="fileFormats:fileFormats triSurface:triSurface surfMesh:surfMesh meshTools:meshTools edgeMesh:edgeMesh
extrudeModel:mesh/extrudeModel dynamicMesh:dynamicMesh lagrangian:lagrangian/basic
decompositionMethods:parallel/decompose/decompositionMethods distributed:parallel/distributed
autoMesh:mesh/autoMesh blockMesh:mesh/blockMesh"

INC="-I$W/inc -I$W/inc_finiteVolume"
for e in $LIBS; do
    n=${e%%:*}; d=${e#*:}
    i="$W/inc_$n"
    if [ ! -f "$i/.done" ]; then
        mkdir -p "$i"
        find "$REF/src/$d" \( -name '*.[CH]' -o -name '*.h' \) -exec ln -sf {} "$i/" \;
        touch "$i/.done"
    fi
    INC="$INC -I$i"
done
CXXFLAGS="-m64 -std=gnu++98 -Dlinux64 -DWM_DP -DNoRepository -ftemplate-depth-100 -O2 -fPIC -w -fpermissive -fno-access-control $INC"

gen_list() {  # $1 = lib dir under src/
    cpp -P -traditional-cpp -DWM_DP -Dlinux64 "$REF/src/$1/Make/files" 2>/dev/null | python3 -c '
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
    if line.startswith("LIB") or line.startswith("EXE") or line.endswith(".H") or line.endswith(".L"): continue
    print(line)'
}

mkdir -p "$W/meshobj"
MK="$W/Makefile.mesh"
{
    echo "CXXFLAGS=$CXXFLAGS"
    echo "ALL="
    for e in $LIBS; do
        n=${e%%:*}; d=${e#*:}
        mkdir -p "$W/meshobj/$n"
        i=0
        objs=""
        for u in $(gen_list "$d"); do
            i=$((i+1))
            o="$W/meshobj/$n/m$i.o"
            objs="$objs $o"
            echo "$o: $REF/src/$d/$u"
            printf '\t@g++ $(CXXFLAGS) -c %s -o %s || echo "FAILED %s" >> %s/meshfailed.txt\n' "$REF/src/$d/$u" "$o" "$d/$u" "$W"
        done
        echo "OBJS_$n=$objs"
        echo "ALL+=\$(OBJS_$n)"
    done
    echo "all: \$(ALL)"
} > "$MK"
rm -f "$W/meshfailed.txt"
make -s -k -f "$MK" -j"$JOBS" all || true
echo "build_ref_mesh.sh: failed units: $(cat "$W/meshfailed.txt" 2>/dev/null | wc -l)"
cat "$W/meshfailed.txt" 2>/dev/null || true

# one shared library per reference library (undefined symbols allowed, as for any .so)
for e in $LIBS; do
    n=${e%%:*}
    g++ -shared -o "$OUT/lib$n.so" "$W"/meshobj/$n/*.o
done
[ -f "$OUT/libfiniteVolume.so" ] || g++ -shared -o "$OUT/libfiniteVolume.so" "$W"/fvobj/*.o

LINK="-Wl,--no-as-needed -L$OUT -lautoMesh -lblockMesh -ldynamicMesh -lextrudeModel -ldecompositionMethods -ldistributed -llagrangian -ledgeMesh -lmeshTools -lsurfMesh -ltriSurface -lfileFormats -lfiniteVolume -lOpenFOAM -ldl -lm -Wl,-rpath,\$ORIGIN -Wl,--allow-shlib-undefined"
APP="$REF/applications/utilities/mesh/generation"
g++ $CXXFLAGS -I"$APP/blockMesh" -c "$APP/blockMesh/blockMeshApp.C" -o "$W/blockMeshApp.o"
g++ -o "$OUT/blockMesh" "$W/blockMeshApp.o" $LINK && echo "build_ref_mesh.sh: OK -> $OUT/blockMesh (the reference's blockMeshApp.C, unchanged)"
g++ $CXXFLAGS -I"$APP/snappyHexMesh" -c "$APP/snappyHexMesh/snappyHexMesh.C" -o "$W/snappyHexMesh.o"
g++ -o "$OUT/snappyHexMesh" "$W/snappyHexMesh.o" $LINK && echo "build_ref_mesh.sh: OK -> $OUT/snappyHexMesh (the reference's snappyHexMesh.C, unchanged)"
# the reference's geometric decomposition methods on a list of cell centres (oracle/decomp_driver.C; decomposePar's `hierarchical`)
g++ $CXXFLAGS -o "$OUT/decomp_driver" "$HERE/decomp_driver.C" $LINK && echo "build_ref_mesh.sh: OK -> $OUT/decomp_driver"
