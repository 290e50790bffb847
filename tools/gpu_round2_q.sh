#!/bin/bash
# round 2, run Q: ticket ring look-ahead sweep
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2q
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_coupled.py -q -m gpu -x > gpurun_out/r2q/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r2q/gpu_tests.log
run() { # name env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu --no-extras --steps 10 > gpurun_out/r2q/bench_$n.json 2> gpurun_out/r2q/bench_$n.err
  python -c "
import json;d=json.load(open('gpurun_out/r2q/bench_$n.json'));print('$n',d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['roofline_vcycle']['frac'])"
}
run pre1_look1 LDU_CLUSTER_PREFETCH=1 LDU_CLUSTER_LOOK=1
run pre1_look2 LDU_CLUSTER_PREFETCH=1 LDU_CLUSTER_LOOK=2
run pre0 LDU_CLUSTER_PREFETCH=0
for ring in 1 0; do for look in 1 2; do
  echo "pcg ring=$ring look=$look"; LDU_CLUSTER_RING=$ring LDU_CLUSTER_LOOK=$look timeout 300 python tools/pcg_probe.py 2>&1 | tail -1
done; done
