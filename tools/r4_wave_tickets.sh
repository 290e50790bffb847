# (round-4 experiment: LDU_WAVE_TICKETS was a knob of a patch that was measured and removed again - DESIGN.md section 7c, "measured and buried" (b); kept for the record of what was run)
mkdir -p gpurun_out/wt
for mesh in motorbike_rcm motorbike; do
i=0
for cfg in "X=0" "LDU_WAVE_TICKETS=1" "LDU_WAVE_TICKETS=1 LDU_SLAB_BPC=2" "LDU_WAVE_TICKETS=1 LDU_SLAB_BPC=3" "LDU_WAVE_TICKETS=1 LDU_P2P_WINDOW=0"; do
  i=$((i+1))
  env $cfg LDU_GAMG_TIME=1 timeout 600 python bench.py --mesh $mesh --steps 1 --warmup 1 --no-extras --no-cpu 2> gpurun_out/wt/${mesh}_$i.err | python -c "
import sys,json
ls=[l for l in sys.stdin if l.startswith('{')]
j=json.loads(ls[-1]) if ls else {'value':None,'ms_per_step':None,'config':{'engine_fallbacks':None}}
print('$mesh | $cfg |', j['value'], j['ms_per_step'], j['config']['engine_fallbacks'])"
  grep "level " gpurun_out/wt/${mesh}_$i.err | tail -19 | awk '{print $(NF-1)}' | tr '\n' ' '
  echo
done; done
