set -u
mkdir -p gpurun_out/r3a
SCOUTER_AUTOTUNE=1 python bench.py --no-cpu-baseline > gpurun_out/r3a/b_auto.json 2> gpurun_out/r3a/b_auto.err
python tools_dev/tune_table.py gpurun_out/r3a/gfx950.json > gpurun_out/r3a/tune.log 2>&1
cp gpurun_out/r3a/gfx950.json scouter_amd/tuning/gfx950.json
python bench.py --no-cpu-baseline > gpurun_out/r3a/b_table.json 2> gpurun_out/r3a/b_table.err
python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r3a/b_table2.json 2>> gpurun_out/r3a/b_table.err
SCOUTER_AUTOTUNE=1 python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r3a/b_auto2.json 2>> gpurun_out/r3a/b_auto.err
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1
tail -5 gpurun_out/r3a/pytest.log
SCOUTER_AUTOTUNE=1 python tools_dev/layer_table.py 70 > gpurun_out/r3a/layer_table.txt 2>&1
python - <<'PY'
import json
for f in ("b_auto","b_table","b_table2","b_auto2"):
    try:
        d=json.loads(open("gpurun_out/r3a/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
