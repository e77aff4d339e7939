mkdir -p gpurun_out/r5m
for v in 0 3 11 15 27 31; do
SCOUTER_X3=$v timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "full_size_resnest26d_224" -s > gpurun_out/r5m/t_$v.txt 2>&1
echo "X3=$v: $(grep -c passed gpurun_out/r5m/t_$v.txt) $(grep -o 'worst |HIP - fp64| / max|grad| = [0-9.e+-]*' gpurun_out/r5m/t_$v.txt | tail -1) $(grep -o '[0-9]* passed\|[0-9]* failed' gpurun_out/r5m/t_$v.txt | tr '\n' ' ')"
done
