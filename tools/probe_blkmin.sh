mkdir -p gpurun_out/r05u
for v in 100 60 30; do
  LDU_BLK_MIN=$v timeout 900 python bench.py --no-cpu --no-extras --no-sublegs --steps 5 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('blkmin=$v', d['value'], d['ms_per_step'], [l[3].split()[0] for l in d['roofline_vcycle']['levels']][-8:])"
done
timeout 2700 python -m pytest tests -q -m gpu -x > gpurun_out/r05u/pytest_gpu.log 2>&1; tail -5 gpurun_out/r05u/pytest_gpu.log
