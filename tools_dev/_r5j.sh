set -x
mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests/test_xslot_gpu.py tests/test_reproducibility_gpu.py tests/test_x3_gpu.py tests/test_graph_gpu.py -x -q > gpurun_out/r5j/test.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5j/test.txt | tail -4
for v in 0 1; do SCOUTER_XSLOT_BWD_SCRATCH=$v python tools_dev/xslot_bench.py 256 300 3 81 3 3 > gpurun_out/r5j/xs81_scratch$v.txt 2>&1; done
python tools_dev/xslot_bench.py 256 300 3 49 3 3 > gpurun_out/r5j/xs49.txt 2>&1
timeout 1500 python tools_dev/tune_x3.py gpurun_out/r5j/gfx950.json > gpurun_out/r5j/tune_x3.txt 2>&1
