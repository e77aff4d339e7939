set -x
mkdir -p gpurun_out/r5h
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q > gpurun_out/r5h/test.txt 2>&1
tail -3 gpurun_out/r5h/test.txt
timeout 1500 python tools_dev/tune_x3.py gpurun_out/r5h/gfx950.json > gpurun_out/r5h/tune_x3.txt 2>&1
cp gpurun_out/r5h/gfx950.json scouter_amd/tuning/gfx950.json
for r in 1 2; do for v in 11 27; do SCOUTER_X3=$v python bench.py --no-cpu-baseline > gpurun_out/r5h/bench_${v}_$r.json 2> gpurun_out/r5h/bench_${v}_$r.err; done; done
