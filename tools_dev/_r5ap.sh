mkdir -p gpurun_out/r5ap
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_storage_gpu.py -x -q -m gpu -k "persistent" > gpurun_out/r5ap/t.txt 2>&1
