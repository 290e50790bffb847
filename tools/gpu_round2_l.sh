#!/bin/bash
mkdir -p gpurun_out/r2l
for cfg in "PROBE_SPIN=200000 LDU_P2P_BPC=2" "PROBE_SPIN=200000 LDU_P2P_BPC=5" "PROBE_SPIN=200000"; do
echo "=== $cfg"; env $cfg LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000 LDU_P2P_SLABS=0 timeout 300 python tools/stuck_probe.py 60 2 2>&1 | grep "fallbacks\|frontier\|^tag"; done
for cfg in "LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000 LDU_P2P_SLABS=0" "LDU_CLUSTER=0 LDU_GS_MAXSKEW=100000" "LDU_CLUSTER=0" "LDU_CLUSTER=2"; do echo "=== $cfg"; env $cfg PROBE_KS=1,2,3,4 timeout 300 python tools/irregular_probe.py 100 2>&1 | grep "^n \|^GS\|^DIC\|rror" | cut -c1-300; done
timeout 600 python bench.py --no-cpu --no-extras > gpurun_out/r2l/bench_box.json 2> gpurun_out/r2l/bench_box.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2l/bench_box.json')); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_vcycle']['frac'])"
