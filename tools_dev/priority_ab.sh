#!/bin/bash
# HIP queue priorities of the step's streams, interleaved A/B of the headline (3 x 60 steps each):
#   base | weight-gradient side stream high | compute stream high | compute + branch high | side + branch high
set -u
O=gpurun_out/prio; mkdir -p $O
B="python bench.py --steps 60 --no-cpu-baseline --no-prof"
for i in 1 2 3; do
  $B 2>/dev/null | tail -1 > $O/base$i.json
  SCOUTER_SIDE_PRIORITY=-1 $B 2>/dev/null | tail -1 > $O/side$i.json
  SCOUTER_MAIN_PRIORITY=-1 $B 2>/dev/null | tail -1 > $O/main$i.json
  SCOUTER_MAIN_PRIORITY=-1 SCOUTER_BRANCH_PRIORITY=-1 $B 2>/dev/null | tail -1 > $O/mainbranch$i.json
  SCOUTER_SIDE_PRIORITY=-1 SCOUTER_BRANCH_PRIORITY=-1 $B 2>/dev/null | tail -1 > $O/sidebranch$i.json
done
python - <<'PY'
import json
for k in ('base','side','main','mainbranch','sidebranch'):
    v=[]
    for i in (1,2,3):
        try: v.append(json.load(open('gpurun_out/prio/%s%d.json'%(k,i)))['value'])
        except Exception as e: v.append(str(e)[:40])
    print(k, v)
PY
