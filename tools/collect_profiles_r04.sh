#!/bin/bash
# Round 4: every file under profiles/r04_* that the numbers in DESIGN.md / README.md are quoted from (run on the GPU box):
#   PMC traffic of the dominant kernels (box, real motorBike mesh in both numberings) - separate --pmc passes;
#   bench lines: box (full: with the real-mesh legs and the extras), the real mesh alone, C5 twin, one rank of 8 (both
#   carriers), 2 and 8 ranks as processes sharing this GPU (functional runs of the N-rank path);
#   rocprofv3 --kernel-trace summaries of the TIMED region of the box and real-mesh benches + dominant kernel durations.
TAG=r04
cd "$(dirname "$0")/.."
R=$PWD
P=$R/profiles
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O /tmp/prof_$TAG
export TMPDIR=/tmp
for spec in box:216 motorbike:mb12:rcm motorbike:mb12; do
  t=${TAG}_$(echo $spec | tr ':' '_')
  timeout 1500 python tools/pmc_traffic.py $t $spec 2 > $O/pmc_$t.log 2>&1; echo "pmc $spec rc=$?"; tail -1 $O/pmc_$t.log | cut -c1-300
done
SECONDS=0
timeout 1800 python bench.py > $P/${TAG}_bench_box.json 2> $O/bench_box.err; echo "bench box rc=$? ($SECONDS s)"
timeout 900 python bench.py --mesh jump2d --n 2000 --no-extras > $P/${TAG}_bench_jump2d.json 2> $O/bench_jump2d.err; echo "bench jump2d rc=$?"
timeout 900 python bench.py --rank-of 8 2> $O/bench_rank8.err | grep '^{' > $P/${TAG}_rank_of_8_projection.json; echo "rank-of 8 rc=$?"
timeout 900 python bench.py --gpus 2 --oversubscribe --steps 3 --no-extras --no-cpu 2> $O/bench_2ranks.err | grep '^{' > $P/${TAG}_bench_2ranks_one_gpu.json; echo "2 ranks rc=$?"
timeout 900 python bench.py --gpus 8 --oversubscribe --n 108 --steps 2 --no-extras --no-cpu --scaling weak 2> $O/bench_8ranks.err | grep '^{' > $P/${TAG}_bench_8ranks_one_gpu_weak108.json; echo "8 ranks rc=$?"
cd /tmp
for m in box motorbike_rcm; do
  LDU_TRACE_MARKER=1 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench_$m -- python $R/bench.py --mesh $m --no-cpu --no-extras --steps 10 > $O/bench_${m}_rocprof.json 2> $O/bench_${m}_rocprof.err; echo "rocprof $m rc=$?"
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_${m}_kernel_trace.csv > $P/${TAG}_bench_${m}_timed_region_kernel_stats.csv
done
{
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_box_kernel_trace.csv --longest sweep_cluster_gs_multi_kernel 40
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_box_kernel_trace.csv --longest "row_kernel<0>" 40
  python -c "
import json;d=json.loads([l for l in open('$O/bench_box_rocprof.json') if l.startswith('{')][-1]);print('box bench under rocprofv3: finest launch avg by HIP events', d['roofline']['avg_launch_ms'], 'ms; Amul', d['amul']['avg_launch_ms'], 'ms; value', d['value'])"
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_motorbike_rcm_kernel_trace.csv --hist sweep_p2p_gs_multi_kernel
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_motorbike_rcm_kernel_trace.csv --hist sweep_slab_gs_multi_kernel
  python -c "
import json;d=json.loads([l for l in open('$O/bench_motorbike_rcm_rocprof.json') if l.startswith('{')][-1]);print('real motorBike mesh (bandCompression numbering) under rocprofv3: finest launch avg by HIP events', d['roofline']['avg_launch_ms'], 'ms; value', d['value'])"
} > $P/${TAG}_dominant_kernel_durations.txt 2>&1
cd $R
cp $P/${TAG}_* $O/ 2>/dev/null
ls -la $P | grep ${TAG}_
