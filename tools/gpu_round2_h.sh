#!/bin/bash
mkdir -p gpurun_out/r2h
for cfg in "LDU_CLUSTER=1" "LDU_CLUSTER=2" "LDU_CLUSTER=0 LDU_P2P_SLABS=0" "LDU_CLUSTER=0 LDU_P2P_SLABS=8" "LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000"; do echo "=== $cfg"; env $cfg LDU_VERBOSE=1 timeout 300 python tools/irregular_probe.py 100 2>&1 | grep -v "amdgpu.ids\|XCD census" | grep "cluster plan\|GS pipeline\|^n \|^GS\|^DIC\|rror" | cut -c1-300; done
echo "=== natural order"; PROBE_RENUMBER=0 LDU_VERBOSE=1 timeout 300 python tools/irregular_probe.py 100 2>&1 | grep "cluster plan\|^n \|^GS\|^DIC\|rror" | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_coupled.py tests/test_gpu_fallback.py tests/test_gpu_scale.py tests/test_gpu_multidomain.py -q -m gpu > gpurun_out/r2h/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2h/tests.log
