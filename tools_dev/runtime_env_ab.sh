#!/bin/bash
# HIP runtime knobs against the headline, interleaved (3 x 60 steps each): kernel arguments in device memory
# (HIP_FORCE_DEV_KERNARG), number of hardware queues the streams are spread over (GPU_MAX_HW_QUEUES)
set -u
O=gpurun_out/rtenv; mkdir -p $O
B="python bench.py --steps 60 --no-cpu-baseline --no-prof"
for i in 1 2 3; do
  $B 2>/dev/null | tail -1 > $O/base$i.json
  HIP_FORCE_DEV_KERNARG=1 $B 2>/dev/null | tail -1 > $O/kernarg1_$i.json
  HIP_FORCE_DEV_KERNARG=0 $B 2>/dev/null | tail -1 > $O/kernarg0_$i.json
  GPU_MAX_HW_QUEUES=8 $B 2>/dev/null | tail -1 > $O/queues8_$i.json
  GPU_MAX_HW_QUEUES=2 $B 2>/dev/null | tail -1 > $O/queues2_$i.json
done
python - <<'PY'
import json
for k in ('base','kernarg1_','kernarg0_','queues8_','queues2_'):
    v=[]
    for i in (1,2,3):
        try:
            d=json.load(open('gpurun_out/rtenv/%s%d.json'%(k,i))); v.append((d['value'], d['config'].get('host_enqueue_ms_per_step')))
        except Exception as e: v.append(str(e)[:40])
    print(k, v)
PY
