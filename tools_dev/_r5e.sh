set -x
mkdir -p gpurun_out/r5e
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_xslot_gpu.py tests/test_engine_gpu.py -x -q > gpurun_out/r5e/test.txt 2>&1
tail -5 gpurun_out/r5e/test.txt
timeout 600 python tools_dev/xwgrad_bench.py 70 > gpurun_out/r5e/xwgrad_bench.txt 2>&1
for v in 3 11; do SCOUTER_X3=$v python bench.py --no-cpu-baseline > gpurun_out/r5e/bench_x3_$v.json 2> gpurun_out/r5e/bench_x3_$v.err; done
