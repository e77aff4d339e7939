mkdir -p gpurun_out/r5ab
timeout 600 python -m pytest tests/test_storage_gpu.py -x -q -m gpu -k "persistent_typed or masked_block" > gpurun_out/r5ab/t.txt 2>&1
timeout 600 python tools_dev/tune_fused_dgrad_bf16.py 256 > gpurun_out/r5ab/tune.txt 2>&1
