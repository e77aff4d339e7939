mkdir -p gpurun_out/r6g
python -m pytest tests/test_xhalo_gpu.py tests/test_x3_gpu.py tests/test_switches_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "Warning\|warn\|^$" | tail -30
python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "fwd_bwd_parity or full_size or config2_at or half_batches" 2>&1 | grep -v "Warning\|warn\|^$" | tail -15
for i in 1 2; do
SCOUTER_X3=63 python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r6g/bench_x63_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r6g/bench_x127_$i.json 2>/dev/null
SCOUTER_X3=255 python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r6g/bench_x255_$i.json 2>/dev/null
done
