#!/bin/bash
mkdir -p gpurun_out/r2g
for cl in 1 2; do echo "=== LDU_CLUSTER=$cl"; LDU_CLUSTER=$cl LDU_VERBOSE=1 timeout 300 python tools/irregular_probe.py 100 2>&1 | grep -v "amdgpu.ids" | tail -12; done
echo "=== natural order (no renumbering)"; PROBE_RENUMBER=0 LDU_VERBOSE=1 timeout 300 python tools/irregular_probe.py 100 2>&1 | grep -v "amdgpu.ids" | tail -8
timeout 1500 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_coupled.py tests/test_gpu_fallback.py tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r2g/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2g/tests.log
timeout 600 python bench.py --no-cpu --no-extras > gpurun_out/r2g/bench_box.json 2> gpurun_out/r2g/bench_box.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2g/bench_box.json')); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_vcycle']['frac'])"
LDU_CLUSTER_NOSEG=1 timeout 600 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NOSEG', d['value'], d['roofline']['avg_launch_ms'])"
LDU_CLUSTER_VARW=1 timeout 600 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('VARW', d['value'], d['roofline']['avg_launch_ms'])"
LDU_GAMG_TIME=1 timeout 300 python tools/gamg_profile.py 216 1 2>&1 | grep "level\|coarsest" | tail -20
