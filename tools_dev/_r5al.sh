mkdir -p gpurun_out/r5al
timeout 900 python -m pytest tests/test_storage_gpu.py -x -q -m gpu > gpurun_out/r5al/t.txt 2>&1
timeout 600 python tools_dev/pwb_dgrad_bench.py 256 > gpurun_out/r5al/f.txt 2>&1
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "config5 or bf16" > gpurun_out/r5al/t2.txt 2>&1
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5al/b5.json 2> gpurun_out/r5al/b5.err
SCOUTER_PWB_DGRAD=0 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5al/b5off.json 2>> gpurun_out/r5al/b5.err
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5al/b5b.json 2>> gpurun_out/r5al/b5.err
SCOUTER_PWB_DGRAD=0 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5al/b5offb.json 2>> gpurun_out/r5al/b5.err
