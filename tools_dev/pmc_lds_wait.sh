export TMPDIR=/tmp
O=${1:-gpurun_out/pmc_lds}; rm -rf $O; mkdir -p $O
B3="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-prof"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  n=$(echo $set | cut -d' ' -f1)
  SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc $set -d $O/$n -- $B3 > $O/$n.log 2>&1
  python tools_dev/pmc_dump.py $(find $O/$n -name "*.db" | head -1) gemm > $O/$n.txt
  python tools_dev/pmc_dump.py $(find $O/$n -name "*.db" | head -1) wgrad >> $O/$n.txt
done
find $O -name "*.db" -delete
cat $O/*.txt
