#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2z
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_coupled.py tests/test_gpu_fullsize.py tests/test_gpu_fallback.py -q -m gpu -x > gpurun_out/r2z/tests.log 2>&1
grep -v amdgpu gpurun_out/r2z/tests.log | tail -60
