set -x
mkdir -p gpurun_out/r5f
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q > gpurun_out/r5f/test.txt 2>&1
tail -3 gpurun_out/r5f/test.txt
SCOUTER_X3_STAGGER=0 timeout 600 python tools_dev/x3_bench.py 70 > gpurun_out/r5f/x3_bench_s0.txt 2>&1
SCOUTER_X3_STAGGER=1 timeout 600 python tools_dev/x3_bench.py 70 > gpurun_out/r5f/x3_bench_s1.txt 2>&1
for r in 1 2; do for v in 0 1; do SCOUTER_X3_STAGGER=$v python bench.py --no-cpu-baseline > gpurun_out/r5f/bench_s${v}_$r.json 2> gpurun_out/r5f/bench_s${v}_$r.err; done; done
