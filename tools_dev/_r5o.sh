mkdir -p gpurun_out/r5o
timeout 1500 python -m pytest tests/test_model_gpu.py -q -k "config5_model" -s > gpurun_out/r5o/t.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5o/t.txt | grep "config 5 @\|passed\|failed\|^E " | head
