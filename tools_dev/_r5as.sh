mkdir -p gpurun_out/r5as
timeout 600 python tools_dev/pwp_plan_bench.py 70 > gpurun_out/r5as/p.txt 2>&1
