#!/bin/bash
# long randomised bit-exact stress of the final round-2 library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2fuzz
export TMPDIR=/tmp
for seed in 31337 4242; do
  timeout 1000 python tools/fuzz_gpu.py 900 $seed > gpurun_out/r2fuzz/fuzz_$seed.log 2>&1; echo "fuzz $seed rc=$?"; tail -1 gpurun_out/r2fuzz/fuzz_$seed.log
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coupled.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
