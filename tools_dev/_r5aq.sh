mkdir -p gpurun_out/r5aq
SCOUTER_X3=47 SCOUTER_HALO=3 timeout 2400 python -m pytest tests/test_model_gpu.py -q -m gpu > gpurun_out/r5aq/t47h.txt 2>&1
SCOUTER_X3=47 timeout 2400 python -m pytest tests/test_model_gpu.py -q -m gpu > gpurun_out/r5aq/t47.txt 2>&1
SCOUTER_HALO=3 timeout 2400 python -m pytest tests/test_model_gpu.py -q -m gpu > gpurun_out/r5aq/th.txt 2>&1
