set -x
mkdir -p gpurun_out/r5a
python bench.py --no-cpu-baseline > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
python tools_dev/conv_bench.py 70 > gpurun_out/r5a/conv_fp32.txt 2>&1
python tools_dev/conv_bench.py 70 bf16 > gpurun_out/r5a/conv_bf16.txt 2>&1
python tools_dev/planes_1x1_bench.py > gpurun_out/r5a/planes_1x1.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r5a/bench2.json 2> gpurun_out/r5a/bench2.err
