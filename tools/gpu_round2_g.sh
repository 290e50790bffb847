#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
LDU_GAMG_TIME=1 timeout 900 python tools/irregular_gamg_probe.py 216 2>&1 | grep "level " | sed -n 33,38p | cut -c1-170
for f in 2 3 4 9; do echo "factor $f"; LDU_SLAB_SEQ_FACTOR=$f timeout 900 python tools/irregular_gamg_probe.py 216 2>&1 | grep "^irregular"; done
LDU_SLAB_SEQ_FACTOR=4 timeout 600 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('box f4',d['value'],d['roofline']['avg_launch_ms'])"
LDU_SLAB_SEQ_FACTOR=9 timeout 600 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('box f9',d['value'],d['roofline']['avg_launch_ms'])"
