#!/bin/bash
# round 2: final validation - full GPU suite, smoke, default bench, mesh variants, torchrun N=1, rocprofv3 summary
cd "$(dirname "$0")/.."
O=gpurun_out/r2final
mkdir -p $O /tmp/prof
export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
/usr/bin/time -v timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
grep -E "Elapsed|Maximum resident" $O/bench_default.err
python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_vcycle']['frac'],d['cpu_baseline'],{k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','unit','cores')}) for k,v in d.get('extras',{}).items()})"
for m in renumbered irregular; do
  timeout 900 python bench.py --mesh $m --no-cpu --no-extras > $O/bench_$m.json 2> $O/bench_$m.err; echo "bench $m rc=$?"
  python -c "
import json;d=json.load(open('$O/bench_$m.json'));print('$m',d['value'],d['roofline']['avg_launch_ms'],d['config'].get('engine_fallbacks'))"
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu --no-extras > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun rc=$?"; cut -c1-200 $O/bench_torchrun1.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --no-cpu --no-extras > $R/$O/bench_rocprof.json 2> $R/$O/bench_rocprof.err; echo "rocprof rc=$?"
cp /tmp/prof/bench_kernel_stats.csv $R/$O/bench_kernel_stats.csv 2>/dev/null || find /tmp/prof -name '*kernel_stats.csv' -exec cp {} $R/$O/bench_kernel_stats.csv \;
head -5 $R/$O/bench_kernel_stats.csv | cut -c1-160
