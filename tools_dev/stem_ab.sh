set -u
mkdir -p gpurun_out/stem
python tools_dev/stem_direct_check.py > gpurun_out/stem/check.log 2>&1
python tools_dev/stem_direct_check.py 70 260 260 >> gpurun_out/stem/check.log 2>&1
timeout 600 python -m pytest tests/test_stem_direct_gpu.py -x -q > gpurun_out/stem/pytest_stem.log 2>&1
for i in 1 2 3; do
  SCOUTER_STEM_DIRECT=0 python bench.py --steps 60 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 > gpurun_out/stem/off$i.json
  SCOUTER_STEM_DIRECT=1 python bench.py --steps 60 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 > gpurun_out/stem/on$i.json
done
python - <<'PY'
import json
for k in ('off','on'):
    print(k, [json.load(open('gpurun_out/stem/%s%d.json'%(k,i)))['value'] for i in (1,2,3)])
PY
tail -3 gpurun_out/stem/check.log; tail -3 gpurun_out/stem/pytest_stem.log
