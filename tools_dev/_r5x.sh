mkdir -p gpurun_out/r5x
python tools_dev/xw_ablate_bench.py > gpurun_out/r5x/ab.txt 2>&1
for n in 1 2 8 16 24 32 128 129 155; do SCOUTER_HIP_LIB=build_dev/libscouter_xwa$n.so python tools_dev/xw_ablate_bench.py >> gpurun_out/r5x/ab.txt 2>&1; done
python tools_dev/xw_ablate_bench.py >> gpurun_out/r5x/ab.txt 2>&1
