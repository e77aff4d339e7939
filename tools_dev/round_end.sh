#!/bin/bash
# One gpurun call that produces everything under profiles/r0N_*: (optionally) the static tile table, the -m gpu suite, every
# rocprofv3 pass (tools_dev/refresh_profiles.sh) and the bench lines of all BASELINE configs + the 260x260 variant.
# usage: bash tools_dev/round_end.sh <outdir> [tune]
set -u
O=${1:-gpurun_out/round_end}
mkdir -p "$O"
if [ "${2:-}" = "tune" ]; then
  python tools_dev/tune_table.py "$O/gfx950.json" > "$O/tune.log" 2>&1
  cp "$O/gfx950.json" scouter_amd/tuning/gfx950.json
fi
# SKIP_TESTS=1: no -m gpu suite; PARTS (refresh_profiles.sh) / CONFIGS pick the configurations, e.g. PARTS=config5 CONFIGS=5
if [ -z "${SKIP_TESTS:-}" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; grep -n "passed\|failed" "$O/pytest.log" | tail -2; fi
bash tools_dev/refresh_profiles.sh "$O/prof" > "$O/refresh.log" 2>&1
# the PMC json files must be in place BEFORE the bench lines are taken (bench.py copies traffic / MFMA-busy from them)
mkdir -p profiles
R=${ROUND:-r06}
cp "$O/prof/pmc_traffic.json" profiles/${R}_pmc_hbm_traffic.json 2>/dev/null
cp "$O/prof/pmc_mfma_util.json" profiles/${R}_pmc_mfma_util.json 2>/dev/null
cp "$O/prof/pmc_traffic_config5.json" profiles/${R}_pmc_hbm_traffic_config5.json 2>/dev/null
cp "$O/prof/pmc_mfma_util_config5.json" profiles/${R}_pmc_mfma_util_config5.json 2>/dev/null
CONFIGS=${CONFIGS:-1 2 3 4 5}
if [[ " $CONFIGS " == *" 2 "* ]]; then
python bench.py --steps 40 2>/dev/null | tail -1 > "$O/bench_line_1gpu.json"
python bench.py --no-cpu-baseline --steps 40 --img-size 260 2>/dev/null | tail -1 > "$O/bench_config2_260.json"
cp "$O/bench_line_1gpu.json" "$O/bench_config2.json"
fi
for c in $CONFIGS; do [ $c = 2 ] || python bench.py --config $c --steps 20 2>/dev/null | tail -1 > "$O/bench_config$c.json"; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    try:
        d = json.load(open(f)); r = d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['dtype'], 'roofline', r.get('frac'), 'traffic', r.get('traffic'))
    except Exception as e:
        print(f, 'ERR', e)
PY
