mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_planes_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "plane or epilogue_finishes" > gpurun_out/r3d/pytest.log 2>&1
tail -15 gpurun_out/r3d/pytest.log
timeout 300 python tools_dev/planes_bench.py > gpurun_out/r3d/planes.txt 2>&1; cat gpurun_out/r3d/planes.txt
