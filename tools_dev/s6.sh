mkdir -p gpurun_out/s6
for rep in 1 2; do for n in 1 2 3; do SCOUTER_SIDE_STREAMS=$n python bench.py --steps 40 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 > gpurun_out/s6/ss${n}_$rep.json; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s6/ss*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'])
PY
