b() { (cd $1 && python bench.py --no-cpu-baseline --no-prof --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"); }
for i in 1 2 3; do b . softmaxfuse; b build_dev/prev colsum1launch; b build_dev/prev2 before_colsum; done
