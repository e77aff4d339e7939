mkdir -p gpurun_out/r5ag
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_table_entries_gpu.py tests/test_kernels_gpu.py tests/test_switches_gpu.py -x -q -m gpu > gpurun_out/r5ag/t.txt 2>&1
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_engine_gpu.py tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r5ag/t2.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5ag/b2.json 2> gpurun_out/r5ag/b2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5ag/b2b.json 2>> gpurun_out/r5ag/b2.err
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5ag/b4.json 2>> gpurun_out/r5ag/b2.err
