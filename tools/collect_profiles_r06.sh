#!/bin/bash
# Round 6: every file under profiles/r05_* that the numbers in DESIGN.md / README.md are quoted from (run on the GPU box):
#   PMC traffic of the dominant kernel of the headline (real motorBike mesh, bandCompression) and of the box - separate --pmc passes;
#   the default bench line (mb12 headline + sub-legs); rocprofv3 --kernel-trace summary of the TIMED region of the headline
#   + dominant kernel durations; one rank of 8 (both carriers); 2 ranks as processes sharing this GPU; PCG leg three times.
TAG=r06
cd "$(dirname "$0")/.."
R=$PWD
P=$R/profiles
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O /tmp/prof_$TAG
export TMPDIR=/tmp
for spec in motorbike:mb12:rcm box:216; do
  t=${TAG}_$(echo $spec | tr ':' '_')
  timeout 1500 python tools/pmc_traffic.py $t $spec 2 > $O/pmc_$t.log 2>&1; echo "pmc $spec rc=$?"; tail -1 $O/pmc_$t.log | cut -c1-300
done
SECONDS=0
timeout 1800 python bench.py > $P/${TAG}_bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? ($SECONDS s)"
timeout 900 python bench.py --mesh box --rank-of 8 2> $O/bench_rank8.err | grep '^{' > $P/${TAG}_rank_of_8_projection.json; echo "rank-of 8 rc=$?"
timeout 900 python bench.py --rank-of 8 --steps 3 --warmup 1 2> $O/bench_rank8_mb12.err | grep '^{' > $P/${TAG}_rank_of_8_projection_mb12.json; echo "rank-of 8 mb12 rc=$?"
timeout 900 python bench.py --mesh box --gpus 2 --oversubscribe --steps 3 --no-extras --no-cpu 2> $O/bench_2ranks.err | grep '^{' > $P/${TAG}_bench_2ranks_one_gpu.json; echo "2 ranks rc=$?"
for i in 1 2 3; do
  timeout 600 python bench.py --mesh box --no-sublegs --no-cpu --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('box run $i: value', d['value'], {k:v for k,v in d['extra'].items() if 'pcg' in k or 'pbicg' in k or 'dic' in k or 'host_pointer' in k})"
done > $P/${TAG}_box_secondary_legs.txt 2>&1
cd /tmp
LDU_TRACE_MARKER=1 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench_mb -- python $R/bench.py --no-cpu --no-extras --steps 10 > $O/bench_mb_rocprof.json 2> $O/bench_mb_rocprof.err; echo "rocprof mb12 rc=$?"
python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_mb_kernel_trace.csv > $P/${TAG}_bench_mb12_rcm_timed_region_kernel_stats.csv
{
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_mb_kernel_trace.csv --hist sweep_p2p_gs_multi_kernel
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_mb_kernel_trace.csv --longest sweep_p2p_gs_multi_kernel 20
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_mb_kernel_trace.csv --hist "gs_blk_kernel<7>"
  python $R/tools/trace_steady.py /tmp/prof_$TAG/bench_mb_kernel_trace.csv --hist "gs_blk_kernel<3>"
  python -c "
import json;d=json.loads([l for l in open('$O/bench_mb_rocprof.json') if l.startswith('{')][-1]);print('real motorBike mesh (bandCompression numbering) under rocprofv3: finest launch avg by HIP events', d['roofline']['avg_launch_ms'], 'ms; value', d['value'])"
} > $P/${TAG}_dominant_kernel_durations.txt 2>&1
cp $O/bench_mb_rocprof.json $P/${TAG}_bench_mb12_rcm_under_rocprof.json
cd $R
cp $P/${TAG}_* $O/ 2>/dev/null     # (gpurun brings back gpurun_out/ only)
ls -la $P | grep ${TAG}_
