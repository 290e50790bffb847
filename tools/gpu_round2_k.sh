#!/bin/bash
for cfg in "PROBE_SPIN=40000000" "PROBE_SPIN=200000 LDU_P2P_BPC=1" "PROBE_SPIN=200000 LDU_P2P_BPC=2" "PROBE_SPIN=200000 LDU_P2P_BPC=5"; do
echo "=== $cfg"; ( time env $cfg LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000 LDU_P2P_SLABS=0 timeout 300 python tools/stuck_probe.py 60 2 2>&1 | grep "fallbacks\|frontier\|^tag" ) 2>&1 | grep -v "^$\|user\|sys"; done
