set -x
mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q > gpurun_out/r5g/test.txt 2>&1
tail -5 gpurun_out/r5g/test.txt
for r in 1 2; do for v in 11 27; do SCOUTER_X3=$v python bench.py --no-cpu-baseline > gpurun_out/r5g/bench_${v}_$r.json 2> gpurun_out/r5g/bench_${v}_$r.err; done; done
