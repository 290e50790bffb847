#!/bin/bash
LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000 LDU_P2P_SLABS=0 LDU_VERBOSE=1 timeout 300 python tools/stuck_probe.py 60 2 2>&1 | grep -v "amdgpu.ids\|XCD census" | tail -40
