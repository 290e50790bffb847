#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2r
export TMPDIR=/tmp
timeout 300 python tools/cluster_trace.py 216 2 > gpurun_out/r2r/trace_216.log 2>&1; echo rc=$?
timeout 300 python tools/cluster_trace.py 108 4 > gpurun_out/r2r/trace_108.log 2>&1; echo rc=$?
cat gpurun_out/r2r/trace_216.log | grep -v amdgpu.ids
