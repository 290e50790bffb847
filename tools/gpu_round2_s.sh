#!/bin/bash
# round 2, run S: abort flag read rarely
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2s
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r2s/gpu_tests.log 2>&1; echo "suite rc=$?"
tail -2 gpurun_out/r2s/gpu_tests.log
run() { n=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu --no-extras --steps 10 > gpurun_out/r2s/bench_$n.json 2> gpurun_out/r2s/bench_$n.err
  python -c "
import json;d=json.load(open('gpurun_out/r2s/bench_$n.json'));print('$n',d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['roofline_vcycle']['frac'])"
}
run bpc3 X=1
run bpc4 LDU_CLUSTER_BPC_MULTI=4
run bpc5 LDU_CLUSTER_BPC_MULTI=5
timeout 300 python tools/pcg_probe.py 2>&1 | tail -1
timeout 300 python tools/pbicg_probe.py 2>&1 | tail -1
timeout 300 python tools/cluster_trace.py 216 2 > gpurun_out/r2s/trace_216.log 2>&1; grep -v amdgpu.ids gpurun_out/r2s/trace_216.log | head -6
timeout 600 python bench.py --mesh irregular --no-cpu --no-extras > gpurun_out/r2s/bench_irregular.json 2> gpurun_out/r2s/bench_irregular.err
python -c "
import json;d=json.load(open('gpurun_out/r2s/bench_irregular.json'));print('irregular',d['value'],d['roofline']['avg_launch_ms'])"
