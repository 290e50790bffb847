#!/bin/bash
mkdir -p gpurun_out/r2f
timeout 900 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2f/tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r2f/tests.log
timeout 600 python bench.py --no-cpu > gpurun_out/r2f/bench_box.json 2> gpurun_out/r2f/bench_box.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2f/bench_box.json')); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline_vcycle']['frac'], d['extra'])"
LDU_GAMG_TIME=1 timeout 300 python tools/gamg_profile.py 216 1 2>&1 | grep "level\|coarsest" | tail -20
LDU_VERBOSE=1 timeout 600 python tools/random_repro.py 1000000 > gpurun_out/r2f/random_repro.log 2>&1; echo "random repro rc=$?"; grep -v "cluster plan\|GAMG level\|addressing:" gpurun_out/r2f/random_repro.log | tail -45
timeout 900 python bench.py --mesh irregular --no-cpu --no-extras > gpurun_out/r2f/bench_irregular.json 2> gpurun_out/r2f/bench_irregular.err; echo "bench irregular rc=$?"; tail -c 2600 gpurun_out/r2f/bench_irregular.json; tail -3 gpurun_out/r2f/bench_irregular.err
timeout 600 python tools/fv_probe.py 216 10 > gpurun_out/r2f/fv_probe.log 2>&1; echo "fv probe rc=$?"; cat gpurun_out/r2f/fv_probe.log | tail -25
