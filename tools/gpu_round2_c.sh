#!/bin/bash
# round-2 check C: GPU suite (continue past failures), GAMG-only kernel + HIP API trace
mkdir -p gpurun_out/r2c
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r2c/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -15 gpurun_out/r2c/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2c/prof -o gamg -- python $GRAFT_REPO_ROOT/tools/gamg_profile.py 216 4 > $GRAFT_REPO_ROOT/gpurun_out/r2c/gamg_profile.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r2c/gamg_profile.log
find gpurun_out/r2c/prof -name "*stats*" | head; for f in $(find gpurun_out/r2c/prof -name "*kernel_stats.csv"); do head -25 $f | cut -c1-160; done
for f in $(find gpurun_out/r2c/prof -name "*hip_api_stats.csv" -o -name "*hip_stats.csv"); do head -20 $f | cut -c1-160; done
for f in $(find gpurun_out/r2c/prof -name "*memory_copy_stats.csv"); do head -10 $f; done
LDU_GAMG_TIME=1 timeout 300 python tools/gamg_profile.py 216 1 > gpurun_out/r2c/gamg_levels.log 2>&1; grep "level\|coarsest" gpurun_out/r2c/gamg_levels.log | tail -45
# keep the traces small
find gpurun_out/r2c/prof -name "*_trace.csv" -size +20M -delete
