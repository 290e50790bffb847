#!/bin/bash
# bench (box, full line), steady-state GAMG trace, unstructured numbers
mkdir -p gpurun_out/r2e
timeout 900 python bench.py > gpurun_out/r2e/bench_box.json 2> gpurun_out/r2e/bench_box.err; echo "bench box rc=$?"; tail -c 3000 gpurun_out/r2e/bench_box.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --selected-regions --output-format csv -d /tmp/prof -o gamg_steady -- python $GRAFT_REPO_ROOT/tools/gamg_profile.py 216 8 > $GRAFT_REPO_ROOT/gpurun_out/r2e/gamg_steady.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
grep "GAMG only" gpurun_out/r2e/gamg_steady.log
cp /tmp/prof/gamg_steady_kernel_stats.csv /tmp/prof/gamg_steady_hip_api_stats.csv gpurun_out/r2e/ 2>/dev/null
head -12 gpurun_out/r2e/gamg_steady_kernel_stats.csv | cut -c1-120
timeout 900 python bench.py --mesh renumbered --no-cpu --no-extras > gpurun_out/r2e/bench_renumbered.json 2> gpurun_out/r2e/bench_renumbered.err; echo "bench renumbered rc=$?"; tail -c 2500 gpurun_out/r2e/bench_renumbered.json; tail -3 gpurun_out/r2e/bench_renumbered.err
timeout 1200 python bench.py --mesh random --no-cpu --no-extras > gpurun_out/r2e/bench_random.json 2> gpurun_out/r2e/bench_random.err; echo "bench random rc=$?"; tail -c 2500 gpurun_out/r2e/bench_random.json; tail -3 gpurun_out/r2e/bench_random.err
