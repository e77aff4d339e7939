mkdir -p gpurun_out/r5ao
SCOUTER_X3=63 timeout 2400 python -m pytest tests/test_model_gpu.py -q -m gpu -s > gpurun_out/r5ao/t63.txt 2>&1
SCOUTER_X3=63 SCOUTER_HALO=3 timeout 2400 python -m pytest tests/test_model_gpu.py -q -m gpu -s > gpurun_out/r5ao/t63h.txt 2>&1
