mkdir -p gpurun_out/r5n
timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/r5n/test_all.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5n/test_all.txt | grep "passed\|failed\|FAILED\|ERROR" | tail -20
