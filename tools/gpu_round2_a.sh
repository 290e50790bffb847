#!/bin/bash
# round-2 check A: the fuzz regression + engine fallback, the whole GPU suite, fuzz with three seeds
mkdir -p gpurun_out/r2a
timeout 600 python -m pytest tests/test_gpu_fallback.py -x -q -m gpu > gpurun_out/r2a/fallback.log 2>&1; echo "fallback rc=$?"; tail -15 gpurun_out/r2a/fallback.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2a/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r2a/gpu_tests.log
for seed in 31337 4242 777; do
  timeout 450 python tools/fuzz_gpu.py 300 $seed > gpurun_out/r2a/fuzz_$seed.log 2>&1; echo "fuzz $seed rc=$?"; tail -2 gpurun_out/r2a/fuzz_$seed.log
done
