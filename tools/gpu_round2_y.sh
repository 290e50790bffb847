#!/bin/bash
# irregular 216^3: where does the per-level time of the level engines go?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2y
export TMPDIR=/tmp
run() { n=$1; shift; echo "== $n"; env "$@" PROBE_KS=1,2,4 timeout 600 python tools/irregular_probe.py 216 2>&1 | grep -E "^GS|^DIC|^n " | cut -c1-200; }
run default X=1
run chipwide LDU_P2P_SLABS=0
run oneslab LDU_P2P_SLABS=1
run win0 LDU_P2P_WINDOW=0
run win32 LDU_P2P_WINDOW=32
run bpc1 LDU_P2P_BPC=1
run bpc4 LDU_P2P_BPC=4
