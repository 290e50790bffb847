#!/bin/bash
# steady-state kernel statistics of the GAMG p-solve (rocprofv3 kernel trace, set-up excluded by tools/trace_steady.py)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2w /tmp/prof
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gamg -- python $R/tools/gamg_profile.py 216 8 > $R/gpurun_out/r2w/gamg_profile.log 2>&1
echo "rocprof rc=$?"
grep "GAMG only" $R/gpurun_out/r2w/gamg_profile.log
find /tmp/prof -type f | head -20
T=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_steady.py "$T" > $R/gpurun_out/r2w/gamg_steady_kernel_stats.csv; echo "steady rc=$?"
cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r2w/gamg_all_kernel_stats.csv
head -30 $R/gpurun_out/r2w/gamg_steady_kernel_stats.csv | cut -c1-150
