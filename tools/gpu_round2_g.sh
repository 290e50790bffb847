#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fallback.py -q -m gpu -x 2>&1 | grep -v amdgpu | tail -4
for w in 1 0; do echo "LDU_GS_WIDE_UPPER=$w"; LDU_GS_WIDE_UPPER=$w timeout 900 python tools/irregular_gamg_probe.py 216 2>&1 | grep "^irregular"; LDU_GS_WIDE_UPPER=$w timeout 900 python tools/irregular_gamg_probe.py 100 2>&1 | grep "^irregular"; done
timeout 600 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('box',d['value'],d['roofline']['avg_launch_ms'])"
timeout 400 python tools/fuzz_gpu.py 200 5150 2>&1 | tail -1
