set -x
mkdir -p gpurun_out/r5d
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5d/test_all.txt 2>&1
tail -15 gpurun_out/r5d/test_all.txt
