timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_reproducibility_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-prof --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['value'], d['ms_per_step'])"; done
