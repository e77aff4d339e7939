mkdir -p gpurun_out/r5at
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_table_entries_gpu.py -x -q -m gpu > gpurun_out/r5at/t.txt 2>&1
timeout 600 python tools_dev/pwp_plan_bench.py 70 > gpurun_out/r5at/p.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5at/b2.json 2> gpurun_out/r5at/b2.err
SCOUTER_PWP_PLAN_DEV=256,1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5at/b2_256.json 2>> gpurun_out/r5at/b2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5at/b2b.json 2>> gpurun_out/r5at/b2.err
SCOUTER_PWP_PLAN_DEV=256,1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5at/b2_256b.json 2>> gpurun_out/r5at/b2.err
