timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_reproducibility_gpu.py -m gpu -x -q 2>&1 | tail -2
b() { (cd $1 && python bench.py --no-cpu-baseline --no-prof --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"); }
for i in 1 2 3; do b . new; b build_dev/prev head; done
