set -x
mkdir -p gpurun_out/r5i
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5i/test_all.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5i/test_all.txt | tail -5
for r in 1 2; do for v in 11 27; do SCOUTER_X3=$v python bench.py --no-cpu-baseline > gpurun_out/r5i/bench_${v}_$r.json 2> gpurun_out/r5i/bench_${v}_$r.err; done; done
