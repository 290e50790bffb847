#!/bin/bash
# TEST INFRASTRUCTURE: oracle/_ref/libdumpSolver.so (oracle/dump_solver.C) against the reference headers collected by
# oracle/build_ref.sh; needs /root/reference at build time only.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
INC="$HERE/_ref/build/inc"
[ -d "$INC" ] || { echo "build_dump_solver.sh: reference headers not available ($INC)" >&2; exit 0; }
g++ -m64 -std=gnu++98 -Dlinux64 -DWM_DP -DNoRepository -ftemplate-depth-100 -O2 -fPIC -w -fpermissive -fno-access-control \
    -I"$INC" -shared -o "$HERE/_ref/libdumpSolver.so" "$HERE/dump_solver.C" -L"$HERE/_ref" -lOpenFOAM -Wl,-rpath,'$ORIGIN'
echo "build_dump_solver.sh: OK -> $HERE/_ref/libdumpSolver.so"
