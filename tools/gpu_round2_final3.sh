#!/bin/bash
# round 2: last validation of the final library
cd "$(dirname "$0")/.."
O=gpurun_out/r2final3
mkdir -p $O /tmp/prof3
export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
SECONDS=0
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? in $SECONDS s"
python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_vcycle']['frac'],d['config']['engine_fallbacks']);print(d['extra'])"
for m in renumbered irregular; do
  timeout 900 python bench.py --mesh $m --no-cpu --no-extras > $O/bench_$m.json 2> $O/bench_$m.err; echo "bench $m rc=$?"
  python -c "
import json;d=json.load(open('$O/bench_$m.json'));print('$m',d['value'],d['roofline']['avg_launch_ms'],d['config'].get('engine_fallbacks'))"
done
cd /tmp
LDU_TRACE_MARKER=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o bench -- python $R/bench.py --no-cpu --no-extras --steps 10 > $R/$O/bench_rocprof.json 2> $R/$O/bench_rocprof.err; echo "rocprof rc=$?"
python $R/tools/trace_steady.py /tmp/prof3/bench_kernel_trace.csv > $R/$O/bench_timed_region_kernel_stats.csv
python $R/tools/trace_steady.py /tmp/prof3/bench_kernel_trace.csv --longest sweep_cluster_gs_multi_kernel 40 > $R/$O/dominant.txt
python $R/tools/trace_steady.py /tmp/prof3/bench_kernel_trace.csv --longest "row_kernel<0>" 40 >> $R/$O/dominant.txt
python -c "
import json;d=json.load(open('$R/$O/bench_rocprof.json'));print('bench under rocprof: finest launch avg', d['roofline']['avg_launch_ms'], 'amul', d['amul']['avg_launch_ms'], 'value', d['value'])" >> $R/$O/dominant.txt
cat $R/$O/dominant.txt | cut -c1-200
timeout 700 python tools/fuzz_gpu.py 300 20260929 > /dev/null 2>&1; cd $R; timeout 700 python tools/fuzz_gpu.py 300 20260929 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
