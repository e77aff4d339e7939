mkdir -p gpurun_out/r3h
python tools_dev/tune_table.py gpurun_out/r3h/gfx950.json > gpurun_out/r3h/tune.log 2>&1
cp gpurun_out/r3h/gfx950.json scouter_amd/tuning/gfx950.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r3h/pytest.log | tail -2
python bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 > gpurun_out/r3h/c2.json
python bench.py --no-cpu-baseline --no-prof --steps 60 --img-size 260 2>/dev/null | tail -1 > gpurun_out/r3h/c2_260.json
for c in 1 3 4 5; do python bench.py --no-cpu-baseline --no-prof --config $c --steps 30 2>/dev/null | tail -1 > gpurun_out/r3h/c$c.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/c*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['dtype'])
PY
