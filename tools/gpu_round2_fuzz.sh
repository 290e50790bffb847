#!/bin/bash
# long randomised bit-exact stress of the final round-2 library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2fuzz
export TMPDIR=/tmp
for seed in 31337 777; do
  timeout 1000 python tools/fuzz_gpu.py 600 $seed > gpurun_out/r2fuzz/fuzz2_$seed.log 2>&1; echo "fuzz $seed rc=$?"; tail -1 gpurun_out/r2fuzz/fuzz2_$seed.log
done
