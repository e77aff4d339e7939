set -x
mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_xslot_gpu.py tests/test_reproducibility_gpu.py -x -q > gpurun_out/r5k/test.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5k/test.txt | tail -4
for v in 0 1; do SCOUTER_XSLOT_BWD_SCRATCH=$v python tools_dev/xslot_bench.py 256 300 3 81 3 3 > gpurun_out/r5k/xs81_scratch$v.txt 2>&1; done
