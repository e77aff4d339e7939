mkdir -p gpurun_out/r5ah
timeout 600 python tools_dev/xpw_fwd_bench.py 70 > gpurun_out/r5ah/f.txt 2>&1
