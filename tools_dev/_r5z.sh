mkdir -p gpurun_out/r5z
python tools_dev/x3_conv_bench.py 70 > gpurun_out/r5z/xc.txt 2>&1
