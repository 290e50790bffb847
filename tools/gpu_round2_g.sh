#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sweep_engines.py -q -m gpu -x 2>&1 | grep -v amdgpu | tail -15
timeout 900 python tools/irregular_gamg_probe.py 216 2>&1 | grep "^irregular"
timeout 600 python bench.py --mesh irregular --no-cpu --no-extras > gpurun_out/bench_irregular_final.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/bench_irregular_final.json'));print('irregular bench',d['value'],d['roofline']['avg_launch_ms'],d['config']['engine_fallbacks'])"
