#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for w in 1 0; do echo "LDU_P2P_WIDE=$w"; LDU_P2P_WIDE=$w timeout 900 python tools/irregular_gamg_probe.py 216 2>&1 | grep "^irregular"; done
