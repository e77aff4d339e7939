export TMPDIR=/tmp
O=gpurun_out/r6gaps; mkdir -p $O
rocprofv3 --kernel-trace -d $O/on -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-prof > $O/on.log 2>&1
DB=$(find $O/on -name "*.db" | head -1)
python tools_dev/gaps.py $DB adamw_kernel:4 > $O/gaps.txt 2>&1
python tools_dev/low_parallel.py $DB adamw_kernel:4 64 > $O/low_parallel.txt 2>&1
find $O -name "*.db" -delete
cat $O/gaps.txt
