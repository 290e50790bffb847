#!/bin/bash
# round 2, run O: split division + no LDS wait between recurrence steps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2o
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r2o/gpu_tests.log 2>&1; echo "suite rc=$?"
tail -3 gpurun_out/r2o/gpu_tests.log
timeout 600 python bench.py --no-cpu --no-extras > gpurun_out/r2o/bench_box.json 2> gpurun_out/r2o/bench_box.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r2o/bench_box.json'));print(d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['roofline_vcycle']['frac'])"
timeout 300 python tools/pcg_probe.py > gpurun_out/r2o/pcg_probe.log 2>&1; tail -5 gpurun_out/r2o/pcg_probe.log
timeout 400 python tools/fuzz_gpu.py 200 4242 > gpurun_out/r2o/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r2o/fuzz.log
