#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -1
for w in 1 0; do echo "LDU_GS_WIDE_UPPER=$w"; LDU_GS_WIDE_UPPER=$w PROBE_KS=1,2,4 timeout 300 python tools/irregular_probe.py 216 2>&1 | grep -E "^GS|^DIC" | tr '\n' ' '; echo; done
timeout 600 python bench.py --no-cpu --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('box',d['value'],d['roofline']['avg_launch_ms'],d['extra'].get('smoothsolver_gs_iterations_per_s'))"
timeout 400 python tools/fuzz_gpu.py 240 2718 2>&1 | tail -1
