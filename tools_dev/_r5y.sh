mkdir -p gpurun_out/r5y
python -m pytest tests/test_x3_gpu.py -x -q -m gpu > gpurun_out/r5y/t.txt 2>&1
rm -f gpurun_out/r5y/ab.txt
for l in "" build_dev/libscouter_xw2set.so "" build_dev/libscouter_xw2set.so; do
  if [ -z "$l" ]; then python tools_dev/xw_ablate_bench.py >> gpurun_out/r5y/ab.txt 2>&1; else SCOUTER_HIP_LIB=$l python tools_dev/xw_ablate_bench.py >> gpurun_out/r5y/ab.txt 2>&1; fi
done
