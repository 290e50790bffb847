#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() { n=$1; shift; echo "== $n"; env "$@" PROBE_KS=1,2,4 timeout 300 python tools/irregular_probe.py 216 2>&1 | grep -E "^GS|^DIC" | tr '\n' ' '; echo; }
run base X=1
run sleep4 LDU_P2P_SLEEP=4
run sleep8 LDU_P2P_SLEEP=8
run bpc2_sleep4 LDU_SLAB_BPC=2 LDU_P2P_SLEEP=4
run bpc2_sleep8 LDU_SLAB_BPC=2 LDU_P2P_SLEEP=8
run bpc2_sleep16 LDU_SLAB_BPC=2 LDU_P2P_SLEEP=16
run sleep1 LDU_P2P_SLEEP=1
