mkdir -p gpurun_out/fresh
python bench.py --steps 40 2>/dev/null | tail -1 > gpurun_out/fresh/bench_line_1gpu.json
python bench.py --no-cpu-baseline --steps 40 --img-size 260 2>/dev/null | tail -1 > gpurun_out/fresh/bench_config2_260.json
python - <<'PY'
import json
for f in ('bench_line_1gpu','bench_config2_260'):
    d=json.load(open('gpurun_out/fresh/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
