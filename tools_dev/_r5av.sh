mkdir -p gpurun_out/r5av
python tools_dev/host_time.py > gpurun_out/r5av/h.txt 2>&1
