mkdir -p gpurun_out/r5ax
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5ax/e1.json 2> gpurun_out/r5ax/err.txt
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5ax/e5.json 2>> gpurun_out/r5ax/err.txt
python tools_dev/host_time.py > gpurun_out/r5ax/h.txt 2>&1
