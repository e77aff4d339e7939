set -x
mkdir -p gpurun_out/r6b
python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "resnest50d_64_spc3 or (other_baseline and resnest50d)" -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r6b/pytest_two.txt
python -m pytest tests/test_model_gpu.py -q -m gpu -k "other_baseline and resnest50d" -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r6b/pytest_other.txt
python tools_dev/head_noise_s300.py 15:2 63:3 31:2 47:2 15:3 > gpurun_out/r6b/head_noise_s300.txt 2>&1
python tools_dev/x3_bench.py 70 > gpurun_out/r6b/x3_bench_70.txt 2>&1
for i in 1 2; do
SCOUTER_X3=15 SCOUTER_HALO=2 python bench.py --config 5 --no-cpu-baseline --no-prof --steps 30 > gpurun_out/r6b/bench5_old_$i.json 2>/dev/null
python bench.py --config 5 --no-cpu-baseline --no-prof --steps 30 > gpurun_out/r6b/bench5_new_$i.json 2>/dev/null
done
python -m pytest tests/test_model_gpu.py tests/test_table_entries_gpu.py -q -m gpu -p no:cacheprovider -k "config1_at or config5_batch256 or plane_forward_entries" 2>&1 | tail -40 > gpurun_out/r6b/pytest_new.txt
