mkdir -p gpurun_out/s5
python bench.py --steps 40 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 > gpurun_out/s5/on1.json
SCOUTER_HEAD_SIDE=0 python bench.py --steps 40 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 > gpurun_out/s5/off1.json
python bench.py --steps 40 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 > gpurun_out/s5/on2.json
SCOUTER_HEAD_SIDE=0 python bench.py --steps 40 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 > gpurun_out/s5/off2.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s5/o*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'])
PY
timeout 2400 python -m pytest tests/test_xslot_gpu.py tests/test_xslot_small_gpu.py tests/test_engine_gpu.py tests/test_graph_gpu.py tests/test_reproducibility_gpu.py tests/test_dp_world2_gpu.py -q -x 2>&1 | tail -4
