mkdir -p gpurun_out/r5p
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-prof --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])"; }
for r in 1 2 3; do
run default X=1
run side0 SCOUTER_SIDE_STREAM=0
run split_sync SCOUTER_SPLIT_ASYNC=0
run branch0 SCOUTER_SIDE_FWD=0 SCOUTER_SIDE_BWD=0
run x3_31 SCOUTER_X3=31
done
python bench.py --no-cpu-baseline --no-prof --steps 40 --graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', d['value'], d['ms_per_step'])"
