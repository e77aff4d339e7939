mkdir -p gpurun_out/r6e
timeout 600 python tools_dev/xhalo_check.py > gpurun_out/r6e/xhalo_check.txt 2>&1
tail -12 gpurun_out/r6e/xhalo_check.txt
for i in 1 2; do
SCOUTER_X3=63 python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r6e/bench_x63_$i.json 2>gpurun_out/r6e/err_x63_$i.txt
python bench.py --no-cpu-baseline --no-prof --steps 60 > gpurun_out/r6e/bench_x127_$i.json 2>gpurun_out/r6e/err_x127_$i.txt
done
python -m pytest tests/test_model_gpu.py -q -m gpu -x -p no:cacheprovider -k "fwd_bwd_parity or full_size" 2>&1 | tail -15
