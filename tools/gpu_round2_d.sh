#!/bin/bash
# GAMG-only kernel + HIP API trace (summaries only come back)
mkdir -p gpurun_out/r2d
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof -o gamg -- python $GRAFT_REPO_ROOT/tools/gamg_profile.py 216 4 > $GRAFT_REPO_ROOT/gpurun_out/r2d/gamg_profile.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/r2d/gamg_profile.log
find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/r2d/ \;
ls -la gpurun_out/r2d; find /tmp/prof -type f | head -20
# which HIP API calls end in a copyBuffer blit: the memory-copy trace (small)
for f in $(find /tmp/prof -name "*memory_copy_trace.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("memory copies:", len(rows), rows[0].keys() if rows else "")
c = collections.Counter()
for r in rows:
    size = int(r.get("Bytes", r.get("bytes", 0)) or 0)
    c[(r.get("Direction", r.get("direction", "?")), size)] += 1
for k, v in c.most_common(25): print(k, v)
PY
done
for f in $(find /tmp/prof -name "*hip_api_trace.csv"); do python - "$f" <<'PY'
import csv, sys, collections
c = collections.Counter(r["Function"] for r in csv.DictReader(open(sys.argv[1])))
for k, v in c.most_common(25): print(k, v)
PY
done
timeout 900 python -m pytest tests/test_gpu_multidomain.py -x -q -m gpu -k "rccl" > gpurun_out/r2d/rccl_tests.log 2>&1; echo "rccl tests rc=$?"; tail -15 gpurun_out/r2d/rccl_tests.log
