#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2x
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sweep_engines.py -q -m gpu -x -k "fused or trace" 2>&1 | tail -2
run() { n=$1; shift
  env "$@" timeout 900 python bench.py --no-cpu --no-extras --steps 10 > gpurun_out/r2x/bench_$n.json 2> gpurun_out/r2x/bench_$n.err
  python -c "
import json;d=json.load(open('gpurun_out/r2x/bench_$n.json'));print('$n',d['value'],d['roofline']['avg_launch_ms'],d['roofline_vcycle']['frac'])"
}
run f0 LDU_FUSE_SMALL=0
run f700 LDU_FUSE_SMALL=700
run f1300 LDU_FUSE_SMALL=1300
run f2560 LDU_FUSE_SMALL=2560
run f5000 LDU_FUSE_SMALL=5000
run f0b LDU_FUSE_SMALL=0
