#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -1
timeout 600 python bench.py --no-cpu --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('box',d['value'],d['roofline']['avg_launch_ms'],d['amul']['avg_launch_ms'],d['amul']['frac'],d['extra'].get('pcg_dic_iterations_per_s'),d['extra'].get('pbicg_dilu_iterations_per_s'))"
timeout 300 python tools/pcg_probe.py 2>&1 | tail -1
