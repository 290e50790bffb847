#!/bin/bash
# randomised bit-exact stress of the final round-2 library (HEAD)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2fuzz
export TMPDIR=/tmp
timeout 1000 python tools/fuzz_gpu.py 600 99 > gpurun_out/r2fuzz/fuzz3_99.log 2>&1; echo "fuzz 99 rc=$?"; tail -1 gpurun_out/r2fuzz/fuzz3_99.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('default bench',d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['config']['engine_fallbacks'])"
