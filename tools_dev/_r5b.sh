set -x
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q > gpurun_out/r5b/test_x3.txt 2>&1
tail -5 gpurun_out/r5b/test_x3.txt
timeout 600 python tools_dev/x3_bench.py 70 > gpurun_out/r5b/x3_bench.txt 2>&1
for v in 0 1 3 7; do SCOUTER_X3=$v python bench.py --no-cpu-baseline > gpurun_out/r5b/bench_x3_$v.json 2> gpurun_out/r5b/bench_x3_$v.err; done
