mkdir -p gpurun_out/r3b
python -m pytest tests/test_model_gpu.py tests/test_engine_gpu.py tests/test_xslot_gpu.py -m gpu -x -q -s --durations=8 > gpurun_out/r3b/pytest.log 2>&1
tail -30 gpurun_out/r3b/pytest.log
