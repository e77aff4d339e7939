mkdir -p gpurun_out/r5ae
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q -m gpu -k "persistent_fused" > gpurun_out/r5ae/t.txt 2>&1
timeout 900 python tools_dev/tune_fused_dgrad_f32.py 70 > gpurun_out/r5ae/tune.txt 2>&1
