#!/bin/bash
mkdir -p gpurun_out/r2n
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r2n/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -6 gpurun_out/r2n/gpu_tests.log
timeout 900 python bench.py --mesh irregular --no-cpu --no-extras > gpurun_out/r2n/bench_irregular.json 2> gpurun_out/r2n/bench_irregular.err; echo "bench irregular rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2n/bench_irregular.json')); print(d['value'], d['ms_per_step'], d['config']['vcycles_per_solve'], d['roofline'], d['roofline_vcycle']['frac'])"; tail -2 gpurun_out/r2n/bench_irregular.err
timeout 900 python bench.py --mesh renumbered --no-cpu --no-extras > gpurun_out/r2n/bench_renumbered.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2n/bench_renumbered.json')); print('renumbered', d['value'], d['roofline_vcycle']['frac'])"
for seed in 31337 99; do timeout 400 python tools/fuzz_gpu.py 240 $seed > gpurun_out/r2n/fuzz_$seed.log 2>&1; echo "fuzz $seed rc=$?"; tail -1 gpurun_out/r2n/fuzz_$seed.log; done
