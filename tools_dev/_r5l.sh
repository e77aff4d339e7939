set -x
bash tools_dev/round_end.sh gpurun_out/r5_end > gpurun_out/r5_end.log 2>&1
tail -20 gpurun_out/r5_end.log
