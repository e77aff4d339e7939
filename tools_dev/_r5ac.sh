mkdir -p gpurun_out/r5ac
timeout 900 python -m pytest tests/test_storage_gpu.py tests/test_table_entries_gpu.py -x -q -m gpu > gpurun_out/r5ac/t.txt 2>&1
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "config5 or bf16" > gpurun_out/r5ac/t2.txt 2>&1
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5ac/b5.json 2> gpurun_out/r5ac/b5.err
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5ac/b5b.json 2>> gpurun_out/r5ac/b5.err
