mkdir -p gpurun_out/r5ar
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r5ar/pytest1.log 2>&1; grep -n "passed\|failed" gpurun_out/r5ar/pytest1.log | tail -2
timeout 1200 python -m pytest tests/test_x3_gpu.py tests/test_storage_gpu.py tests/test_table_entries_gpu.py -m gpu -q > gpurun_out/r5ar/pytest2.log 2>&1; grep -n "passed\|failed" gpurun_out/r5ar/pytest2.log | tail -2
timeout 1200 python -m pytest tests/test_x3_gpu.py tests/test_storage_gpu.py tests/test_table_entries_gpu.py -m gpu -q > gpurun_out/r5ar/pytest3.log 2>&1; grep -n "passed\|failed" gpurun_out/r5ar/pytest3.log | tail -2
