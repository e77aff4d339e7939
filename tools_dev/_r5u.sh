mkdir -p gpurun_out/r5u
for r in 1 2 3; do
python tools_dev/x3_ablate_bench.py >> gpurun_out/r5u/ab.txt 2>&1
SCOUTER_HIP_LIB=build_dev/libscouter_x3prio.so python tools_dev/x3_ablate_bench.py >> gpurun_out/r5u/ab.txt 2>&1
done
