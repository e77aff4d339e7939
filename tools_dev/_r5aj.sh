mkdir -p gpurun_out/r5aj
timeout 900 python -m pytest tests/test_storage_gpu.py -x -q -m gpu > gpurun_out/r5aj/t.txt 2>&1
timeout 600 python tools_dev/pwb_fwd_bench.py 256 > gpurun_out/r5aj/f.txt 2>&1
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "config5 or bf16" > gpurun_out/r5aj/t2.txt 2>&1
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5aj/b5.json 2> gpurun_out/r5aj/b5.err
SCOUTER_PWB_FWD=0 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5aj/b5off.json 2>> gpurun_out/r5aj/b5.err
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5aj/b5b.json 2>> gpurun_out/r5aj/b5.err
SCOUTER_PWB_FWD=0 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5aj/b5offb.json 2>> gpurun_out/r5aj/b5.err
