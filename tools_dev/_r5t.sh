mkdir -p gpurun_out/r5t
python tools_dev/x3_ablate_bench.py > gpurun_out/r5t/ab.txt 2>&1
for n in 1 2 4 6 7 8 16 24 31 32 63 64 95; do SCOUTER_HIP_LIB=build_dev/libscouter_x3a$n.so python tools_dev/x3_ablate_bench.py >> gpurun_out/r5t/ab.txt 2>&1; done
python tools_dev/x3_ablate_bench.py >> gpurun_out/r5t/ab.txt 2>&1
