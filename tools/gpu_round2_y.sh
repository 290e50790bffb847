#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() { n=$1; shift; echo "== $n"; env "$@" PROBE_KS=1,3,4 timeout 300 python tools/irregular_probe.py 216 2>&1 | grep -E "^GS|^DIC" | tr '\n' ' '; echo; }
run auto X=1
run bpc1 LDU_SLAB_BPC=1
run bpc2 LDU_SLAB_BPC=2
run bpc3 LDU_SLAB_BPC=3
run bpc4 LDU_SLAB_BPC=4
run bpc1_win16 LDU_SLAB_BPC=1 LDU_P2P_WINDOW=16
