set -x
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_planes_gpu.py -x -q > gpurun_out/r5c/test_x3.txt 2>&1
tail -5 gpurun_out/r5c/test_x3.txt
timeout 1200 python tools_dev/tune_x3.py gpurun_out/r5c/gfx950.json > gpurun_out/r5c/tune_x3.txt 2>&1
tail -3 gpurun_out/r5c/tune_x3.txt
cp gpurun_out/r5c/gfx950.json scouter_amd/tuning/gfx950.json
for v in 0 3; do for a in 1 0; do SCOUTER_SPLIT_ASYNC=$a SCOUTER_X3=$v python bench.py --no-cpu-baseline > gpurun_out/r5c/bench_x3_${v}_a$a.json 2> gpurun_out/r5c/bench_x3_${v}_a$a.err; done; done
SCOUTER_X3=3 python bench.py --no-cpu-baseline > gpurun_out/r5c/bench_x3_3_again.json 2>/dev/null
