# SQ-side counters of ONE convolution layer: where the wave cycles go (issue stalls vs waits vs VALU vs MFMA busy).
# usage: bash tools_dev/pmc_one_conv.sh <outfile> MODE cin cout k groups H [B] [tile]
export TMPDIR=/tmp
OUT=$1; shift
D=/tmp/pmc_one_$$; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $D -- python tools_dev/one_conv.py "$@" > $D/log 2>&1
echo "== $@" >> $OUT
python tools_dev/pmc_dump.py $(find $D -name "*.db" | head -1) gemm >> $OUT
python tools_dev/pmc_dump.py $(find $D -name "*.db" | head -1) wgrad >> $OUT
python tools_dev/pmc_dump.py $(find $D -name "*.db" | head -1) pconv >> $OUT
python tools_dev/pmc_dump.py $(find $D -name "*.db" | head -1) pwgrad >> $OUT
python tools_dev/pmc_dump.py $(find $D -name "*.db" | head -1) phalo >> $OUT
if [ -n "$PMC2" ]; then
  rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS -d $D -- python tools_dev/one_conv.py "$@" > $D/log 2>&1
  python tools_dev/pmc_dump.py $(find $D -name "*.db" | head -1) pconv >> $OUT
  python tools_dev/pmc_dump.py $(find $D -name "*.db" | head -1) pwgrad >> $OUT
  python tools_dev/pmc_dump.py $(find $D -name "*.db" | head -1) phalo >> $OUT
fi
rm -rf $D
