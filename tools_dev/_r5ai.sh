mkdir -p gpurun_out/r5ai
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r5ai/t.txt 2>&1
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_switches_gpu.py tests/test_engine_gpu.py tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r5ai/t2.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5ai/b2.json 2> gpurun_out/r5ai/b2.err
SCOUTER_X3=15 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5ai/b2_15.json 2>> gpurun_out/r5ai/b2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5ai/b2b.json 2>> gpurun_out/r5ai/b2.err
SCOUTER_X3=15 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5ai/b2_15b.json 2>> gpurun_out/r5ai/b2.err
