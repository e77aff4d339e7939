mkdir -p gpurun_out/s1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_xslot_small_gpu.py -q 2>&1 | tail -30 > gpurun_out/s1/test_small.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -- python tools_dev/xslot_bench.py 70 10 1 49 3 3 > /tmp/prof_s.log 2>&1
python tools_dev/rocpd_summary.py "$(find /tmp/prof_s -name '*.db' | head -1)" > gpurun_out/s1/rocprof_small.txt 2>&1
cat gpurun_out/s1/test_small.txt gpurun_out/s1/rocprof_small.txt
