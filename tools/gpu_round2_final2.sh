#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2final
mkdir -p $O /tmp/prof2
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_gpu_sweep_engines.py tests/test_gpu_parity.py tests/test_gpu_coupled.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -2
SECONDS=0
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? in $SECONDS s"
python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_vcycle']['frac'],d['cpu_baseline']);print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','unit','cores','error')}) for k,v in d.get('extras',{}).items()})"
cd /tmp
LDU_TRACE_MARKER=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o bench -- python $R/bench.py --no-cpu --no-extras --steps 10 > $R/$O/bench_rocprof.json 2> $R/$O/bench_rocprof.err; echo "rocprof rc=$?"
python $R/tools/trace_steady.py /tmp/prof2/bench_kernel_trace.csv > $R/$O/bench_timed_region_kernel_stats.csv; echo "steady rc=$?"
cp /tmp/prof2/bench_kernel_stats.csv $R/$O/bench_whole_run_kernel_stats.csv
head -4 $R/$O/bench_timed_region_kernel_stats.csv | cut -c1-120
grep -c . $R/$O/bench_timed_region_kernel_stats.csv
