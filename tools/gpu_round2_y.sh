#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() { n=$1; shift; echo "== $n"; env "$@" PROBE_KS=1,2,3,4 timeout 300 python tools/irregular_probe.py 216 2>&1 | grep -E "^GS|^DIC" | tr '\n' ' '; echo; }
run pairs X=1
timeout 900 python tools/irregular_gamg_probe.py 216 2>&1 | grep "^irregular"
LDU_GS_PAIRSKEW=1000000 timeout 900 python tools/irregular_gamg_probe.py 216 2>&1 | grep "^irregular"
timeout 900 python tools/irregular_gamg_probe.py 100 2>&1 | grep "^irregular"
timeout 600 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('box',d['value'],d['roofline']['avg_launch_ms'])"
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -1
