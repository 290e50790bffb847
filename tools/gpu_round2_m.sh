#!/bin/bash
mkdir -p gpurun_out/r2m
for cfg in "PROBE_SPIN=200000 LDU_P2P_BPC=5" "PROBE_SPIN=200000"; do
echo "=== $cfg"; env $cfg LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000 LDU_P2P_SLABS=0 timeout 300 python tools/stuck_probe.py 60 2 2>&1 | grep "fallbacks\|frontier\|^tag"; done
for cfg in "LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000 LDU_P2P_SLABS=0" "LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000" "LDU_CLUSTER=0" "LDU_CLUSTER=0 LDU_P2P_WINDOW=0" "LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000 LDU_P2P_WINDOW=4" "LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000 LDU_P2P_WINDOW=16"; do echo "=== $cfg"; env $cfg PROBE_KS=1,2,3,4 timeout 300 python tools/irregular_probe.py 100 2>&1 | grep "^n \|^GS\|^DIC\|rror" | cut -c1-300; done
timeout 600 python bench.py --no-cpu --no-extras > gpurun_out/r2m/bench_box.json 2> gpurun_out/r2m/bench_box.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2m/bench_box.json')); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_vcycle']['frac'])"
timeout 1500 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fallback.py -x -q -m gpu > gpurun_out/r2m/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r2m/tests.log
