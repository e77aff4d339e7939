mkdir -p gpurun_out/r5af
timeout 600 python tools_dev/xpw_debug.py > gpurun_out/r5af/d.txt 2>&1
