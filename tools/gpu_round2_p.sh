#!/bin/bash
# round 2, run P: ticket ring + software-pipelined cluster GaussSeidel sweeps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2p
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fallback.py tests/test_gpu_scale.py -q -m gpu -x > gpurun_out/r2p/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2p/gpu_tests.log
for pre in 1 0; do
LDU_CLUSTER_PREFETCH=$pre timeout 600 python bench.py --no-cpu --no-extras > gpurun_out/r2p/bench_box_pre$pre.json 2> gpurun_out/r2p/bench_box_pre$pre.err; echo "bench pre=$pre rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r2p/bench_box_pre$pre.json'));print(d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['roofline_vcycle']['frac'])"
done
timeout 400 python tools/fuzz_gpu.py 150 777 > gpurun_out/r2p/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r2p/fuzz.log
