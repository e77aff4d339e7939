mkdir -p gpurun_out/r5ay
python tools_dev/host_profile.py > gpurun_out/r5ay/p.txt 2>&1
