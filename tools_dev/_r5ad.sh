mkdir -p gpurun_out/r5ad
timeout 600 python tools_dev/pwb_debug.py > gpurun_out/r5ad/d2.txt 2>&1
timeout 900 python -m pytest tests/test_storage_gpu.py tests/test_table_entries_gpu.py -x -q -m gpu > gpurun_out/r5ad/t.txt 2>&1
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "config5 or bf16" > gpurun_out/r5ad/t2.txt 2>&1
timeout 600 python tools_dev/tune_fused_dgrad_bf16.py 256 > gpurun_out/r5ad/tune.txt 2>&1
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5ad/b5.json 2> gpurun_out/r5ad/b5.err
