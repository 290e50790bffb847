#!/bin/bash
# round-2 check B: whole GPU suite incl. the new fallback / nonorth / pitzDaily / drop-in tests, fuzz replay across case 8723
mkdir -p gpurun_out/r2b
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -25 gpurun_out/r2b/gpu_tests.log
FUZZ_START=8700 timeout 900 python tools/fuzz_gpu.py 900 31337 > gpurun_out/r2b/fuzz_replay.log 2>&1; echo "fuzz replay rc=$?"; tail -3 gpurun_out/r2b/fuzz_replay.log
