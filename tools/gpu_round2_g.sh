#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r2final4
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
timeout 900 python bench.py --mesh irregular --no-cpu --no-extras > gpurun_out/r2final4/bench_irregular.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2final4/bench_irregular.json'));print('irregular bench',d['value'],d['roofline']['avg_launch_ms'],d['config']['engine_fallbacks'])"
timeout 900 python bench.py --mesh renumbered --no-cpu --no-extras > gpurun_out/r2final4/bench_renumbered.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2final4/bench_renumbered.json'));print('renumbered bench',d['value'],d['roofline']['avg_launch_ms'],d['config']['engine_fallbacks'])"
timeout 900 python bench.py > gpurun_out/r2final4/bench_default.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2final4/bench_default.json'));print('default bench',d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['roofline_vcycle']['frac'],d['extra'])"
timeout 400 python tools/fuzz_gpu.py 300 8086 2>&1 | tail -1
