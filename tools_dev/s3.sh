mkdir -p gpurun_out/s3
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_h -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-prof > /tmp/prof_h.log 2>&1
python tools_dev/head_section.py "$(find /tmp/prof_h -name '*.db' | head -1)" 14 36 > gpurun_out/s3/head_section.txt 2>&1
cat gpurun_out/s3/head_section.txt
