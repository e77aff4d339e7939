mkdir -p gpurun_out/s4
timeout 1500 python -m pytest tests/test_xslot_gpu.py tests/test_xslot_small_gpu.py tests/test_reproducibility_gpu.py tests/test_model_gpu.py -q -x 2>&1 | tail -6 > gpurun_out/s4/tests.txt
cat gpurun_out/s4/tests.txt
