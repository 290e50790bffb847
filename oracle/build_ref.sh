#!/bin/bash
# TEST INFRASTRUCTURE — builds the *reference's own* libOpenFOAM (serial, dummy
# Pstream) from the sources where they lie under /root/reference, bypassing
# wmake (needs flex, absent here).  Recipe = SURVEY.md Appendix A.
#
# Outputs ONLY into oracle/_ref/ (git-ignored; travels to the GPU box as a
# binary).  Nothing from the reference is copied into git history.
#   oracle/_ref/libOpenFOAM.so     the reference library (411 units + POSIX + dummy Pstream)
#   oracle/_ref/ref_driver         oracle/ref_driver.C linked against it
#   oracle/_ref/etc/controlDict    run-time config the library reads at static-init
#   oracle/_ref/build/             scratch objects (listed in .gpurunignore)
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
W="$OUT/build"
JOBS=${JOBS:-8}

if [ ! -d "$REF/src/OpenFOAM" ]; then
    echo "build_ref.sh: $REF not present (GPU box?) - using prebuilt oracle/_ref" >&2
    exit 0
fi

mkdir -p "$W/inc" "$W/obj" "$OUT/etc"

# 1. flat include dir (what wmakeLnInclude does)
if [ ! -f "$W/inc/.done" ]; then
    for d in src/OpenFOAM src/OSspecific/POSIX; do
        find "$REF/$d" \( -name '*.[CH]' -o -name '*.h' \) -exec ln -sf {} "$W/inc/" \;
    done
    touch "$W/inc/.done"
fi

# 2. source lists from the cpp-preprocessed Make/files
gen_list() {  # $1 = lib dir under src/
    cpp -P -traditional-cpp -DWM_DP -Dlinux64 "$REF/src/$1/Make/files" 2>/dev/null | python3 -c '
import sys,re
vars={}
out=[]
for line in sys.stdin:
    line=line.strip()
    if not line: continue
    m=re.match(r"^(\w+)\s*=\s*(.*)$", line)
    if m:
        v=m.group(2)
        for k,val in vars.items(): v=v.replace("$(%s)"%k,val)
        vars[m.group(1)]=v; continue
    for k,val in vars.items(): line=line.replace("$(%s)"%k,val)
    if line.startswith("LIB") or line.startswith("EXE"): continue
    out.append(line)
print("\n".join(out))'
}

{
    gen_list OpenFOAM | sed "s!^!$REF/src/OpenFOAM/!"
    gen_list OSspecific/POSIX | sed "s!^!$REF/src/OSspecific/POSIX/!"
    for f in UPstream.C UIPread.C UOPwrite.C; do echo "$REF/src/Pstream/dummy/$f"; done
} > "$W/sources.txt"

# global.Cver needs its version strings substituted (wmake/rules/General/version)
sed -e 's!VERSION_STRING!2.2.x!' -e 's!BUILD_STRING!oracle!' \
    "$REF/src/OpenFOAM/global/global.Cver" > "$W/global.C"

CXXFLAGS="-m64 -std=gnu++98 -Dlinux64 -DWM_DP -DNoRepository -ftemplate-depth-100 -O2 -fPIC -w -fpermissive -fno-access-control -I$W/inc"

# 3. Makefile over all units
{
    echo "CXXFLAGS=$CXXFLAGS"
    echo "OBJS="
    i=0
    while read -r src; do
        i=$((i+1))
        case "$src" in *.H) continue ;; esac   # Make/files lists one stray header
        case "$src" in
            *global.Cver) s="$W/global.C"; extra="-I$REF/src/OpenFOAM/global" ;;
            *sigFpe.C)    s="$src"; extra="-U__linux__ -Ulinux -U__linux" ;;
            *)            s="$src"; extra="" ;;
        esac
        o="$W/obj/u$i.o"
        echo "OBJS+=$o"
        echo "$o: $s"
        printf '\t@g++ $(CXXFLAGS) %s -c %s -o %s\n' "$extra" "$s" "$o"
    done < "$W/sources.txt"
    echo "all: \$(OBJS)"
} > "$W/Makefile"

if [ -n "$DRIVERS_ONLY" ] && [ -f "$OUT/libOpenFOAM.so" ]; then
    # re-link only the driver (our code) against the reference library already built
    g++ $CXXFLAGS -o "$OUT/ref_driver" "$HERE/ref_driver.C" -L"$OUT" -lOpenFOAM -ldl -lm -Wl,-rpath,'$ORIGIN'
    echo "build_ref.sh: OK (driver only) -> $OUT/ref_driver"
    exit 0
fi
make -s -C "$W" -f "$W/Makefile" -j"$JOBS" all

# 4. link
g++ -shared -o "$OUT/libOpenFOAM.so" "$W"/obj/*.o -ldl -lz -lm

# 5. run-time config: the library looks up etc/controlDict under $WM_PROJECT_DIR
cp "$REF/etc/controlDict" "$REF/etc/cellModels" "$OUT/etc/"

# 6. the driver (our code, reference headers)
[ -f "$HERE/ref_driver.C" ] && g++ $CXXFLAGS -o "$OUT/ref_driver" "$HERE/ref_driver.C" -L"$OUT" -lOpenFOAM -ldl -lm \
    -Wl,-rpath,'$ORIGIN'
echo "build_ref.sh: OK -> $OUT/libOpenFOAM.so, $OUT/ref_driver"
