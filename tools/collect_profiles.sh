#!/bin/bash
# Produces every file under profiles/<TAG>_* that bench.py's numbers are judged on (run on the GPU box through gpurun):
#   bench lines (box = the metric's workload, octree twin in both numberings, C5 twin, single-rank projection of 8 ranks),
#   rocprofv3 --kernel-trace summaries of the TIMED region of the box and octree benches + the dominant kernels' durations,
#   PMC traffic of the dominant kernels (separate --pmc passes, tools/pmc_traffic.py).
# usage: bash tools/collect_profiles.sh r03        (outputs also copied to gpurun_out/<TAG>_profiles/)
TAG=${1:-r03}
cd "$(dirname "$0")/.."
R=$PWD
P=$R/profiles
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O /tmp/prof_$TAG
export TMPDIR=/tmp
run() { echo "== $*"; "$@"; }
timeout 1500 python tools/pmc_traffic.py ${TAG}_box box:216 2 > $O/pmc_box.log 2>&1; echo "pmc box rc=$?"; tail -1 $O/pmc_box.log | cut -c1-300
timeout 1500 python tools/pmc_traffic.py ${TAG}_octree octree:14:6:7 2 > $O/pmc_octree.log 2>&1; echo "pmc octree rc=$?"; tail -1 $O/pmc_octree.log | cut -c1-300
SECONDS=0
timeout 1500 python bench.py > $P/${TAG}_bench_box.json 2> $O/bench_box.err; echo "bench box rc=$? ($SECONDS s)"
for m in octree octree_hexref; do
  timeout 1200 python bench.py --mesh $m --no-extras > $P/${TAG}_bench_$m.json 2> $O/bench_$m.err; echo "bench $m rc=$?"
done
timeout 900 python bench.py --mesh jump2d --n 2000 --no-extras > $P/${TAG}_bench_jump2d.json 2> $O/bench_jump2d.err; echo "bench jump2d rc=$?"
timeout 900 python bench.py --mesh renumbered --no-cpu --no-extras > $P/${TAG}_bench_renumbered.json 2> $O/bench_renumbered.err; echo "bench renumbered rc=$?"
timeout 900 python bench.py --mesh irregular --no-cpu --no-extras > $P/${TAG}_bench_irregular.json 2> $O/bench_irregular.err; echo "bench irregular rc=$?"
{ echo '```'; timeout 900 python tools/fv_probe.py 216 20 2>&1 | grep -v amdgpu.ids; echo '```'; } > $P/${TAG}_fv_probe.md; echo "fv probe rc=$?"
timeout 900 python bench.py --rank-of 8 2> $O/bench_rank8.err | grep '^{' > $P/${TAG}_rank_of_8_projection.json; echo "rank-of 8 rc=$?"
cd /tmp
for m in box octree; do
  LDU_TRACE_MARKER=1 timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench_$m -- python $R/bench.py --mesh $m --no-cpu --no-extras --steps 10 > $O/bench_${m}_rocprof.json 2> $O/bench_${m}_rocprof.err; echo "rocprof $m rc=$?"
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_${m}_kernel_trace.csv > $P/${TAG}_bench_${m}_timed_region_kernel_stats.csv
done
{
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_box_kernel_trace.csv --longest sweep_cluster_gs_multi_kernel 40
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_box_kernel_trace.csv --longest "row_kernel<0>" 40
  python -c "
import json;d=json.load(open('$O/bench_box_rocprof.json'));print('box bench under rocprofv3: finest launch avg by HIP events', d['roofline']['avg_launch_ms'], 'ms; Amul', d['amul']['avg_launch_ms'], 'ms; value', d['value'])"
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_octree_kernel_trace.csv --hist sweep_p2p_gs_multi_kernel
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_octree_kernel_trace.csv --hist sweep_slab_gs_multi_kernel
  python -c "
import json;d=json.load(open('$O/bench_octree_rocprof.json'));print('octree bench under rocprofv3: finest launch avg by HIP events', d['roofline']['avg_launch_ms'], 'ms; value', d['value'])"
} > $P/${TAG}_dominant_kernel_durations.txt 2>&1
cd $R
cp $P/${TAG}_* $O/ 2>/dev/null
ls -la $P | grep ${TAG}_
