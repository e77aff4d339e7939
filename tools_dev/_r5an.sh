mkdir -p gpurun_out/r5an
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -k "rounding_noise" -s > gpurun_out/r5an/t.txt 2>&1
