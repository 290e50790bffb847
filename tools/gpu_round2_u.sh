#!/bin/bash
# round 2, run U: 8-wide polls for wide rows in the level engines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2u
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r2u/gpu_tests.log 2>&1; echo "suite rc=$?"
tail -2 gpurun_out/r2u/gpu_tests.log
run() { n=$1; shift
  timeout 900 python bench.py --no-cpu --no-extras "$@" > gpurun_out/r2u/bench_$n.json 2> gpurun_out/r2u/bench_$n.err
  python -c "
import json;d=json.load(open('gpurun_out/r2u/bench_$n.json'));print('$n',d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['roofline_vcycle']['frac'])"
}
run box
run irregular --mesh irregular
timeout 600 python tools/irregular_probe.py 100 > gpurun_out/r2u/irregular_probe.log 2>&1; tail -12 gpurun_out/r2u/irregular_probe.log
timeout 400 python tools/fuzz_gpu.py 200 99 > gpurun_out/r2u/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r2u/fuzz.log
