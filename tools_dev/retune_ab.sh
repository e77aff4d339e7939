#!/bin/bash
# Re-makes the static tile table on this box (tools_dev/tune_table.py) and A/Bs it against the committed one on the headline
# config and config 5, interleaved.  usage: bash tools_dev/retune_ab.sh [outdir]  -> <outdir>/gfx950.json, ab.txt
O=${1:-gpurun_out/retune}
mkdir -p "$O"
python tools_dev/tune_table.py "$O/gfx950.json" > "$O/tune.log" 2>&1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  python bench.py --steps 40 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 | line "config2 committed"
  SCOUTER_TUNE_TABLE="$O/gfx950.json" python bench.py --steps 40 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 | line "config2 retuned"
done 2>&1 | tee "$O/ab.txt"
for i in 1 2; do
  python bench.py --config 5 --steps 20 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 | line "config5 committed"
  SCOUTER_TUNE_TABLE="$O/gfx950.json" python bench.py --config 5 --steps 20 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 | line "config5 retuned"
done 2>&1 | tee -a "$O/ab.txt"
tail -3 "$O/tune.log"
