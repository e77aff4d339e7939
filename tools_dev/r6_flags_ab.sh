set -x
mkdir -p gpurun_out/r6a
python tools_dev/seed_noise.py > gpurun_out/r6a/seed_noise.txt 2>&1
SCOUTER_X3=63 SCOUTER_HALO=3 timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/r6a/pytest_flipped.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r6a/bench_default_$i.json 2>/dev/null
SCOUTER_X3=63 SCOUTER_HALO=3 python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r6a/bench_flipped_$i.json 2>/dev/null
done
