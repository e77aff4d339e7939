#!/bin/bash
# Runs on the GPU box (gpurun): every rocprofv3 pass behind profiles/r01_*, summarised in place (the rocpd databases
# are too big to travel back).  usage: bash tools_dev/refresh_profiles.sh [outdir]   -> <outdir>/*.txt, *.json
set -u
export TMPDIR=/tmp
O=${1:-gpurun_out/prof_refresh}
mkdir -p "$O"
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof"
B3="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-prof"
db() { find "$1" -name "*.db" | head -1; }
PARTS=${PARTS:-main config5}          # PARTS=main: the headline config's passes only; PARTS=config5: BASELINE configs[4]'s only
if [[ " $PARTS " == *" main "* ]]; then
SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d "$O/off" -- $B > "$O/off.log" 2>&1
python tools_dev/rocpd_summary.py "$(db $O/off)" adamw_kernel:3 > "$O/sum_off.txt"
rocprofv3 --kernel-trace --stats -d "$O/on" -- $B > "$O/on.log" 2>&1
python tools_dev/rocpd_summary.py "$(db $O/on)" adamw_kernel:3 > "$O/sum_on.txt"
rocprofv3 --kernel-trace --stats -d "$O/xs" -- python tools_dev/xslot_bench.py 256 300 3 49 3 3 > "$O/xs.log" 2>&1
python tools_dev/rocpd_summary.py "$(db $O/xs)" > "$O/sum_xs.txt"
rocprofv3 --kernel-trace --stats -d "$O/xsh" -- python tools_dev/xslot_bench.py 70 10 1 49 3 3 > "$O/xsh.log" 2>&1
python tools_dev/rocpd_summary.py "$(db $O/xsh)" > "$O/sum_xs_head.txt"
rocprofv3 --kernel-trace --stats -d "$O/xs81" -- python tools_dev/xslot_bench.py 256 300 3 81 3 3 > "$O/xs81.log" 2>&1
python tools_dev/rocpd_summary.py "$(db $O/xs81)" > "$O/sum_xs81.txt"
SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/fetch" -- $B3 > "$O/fetch.log" 2>&1
SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/write" -- $B3 > "$O/write.log" 2>&1
python tools_dev/pmc_traffic.py "$(db $O/fetch)" "$(db $O/write)" adamw_kernel:3 "$O/pmc_traffic.json" 3 > "$O/pmc_traffic.txt" 2> "$O/pmc_traffic.err"
SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d "$O/mfma" -- $B3 > "$O/mfma.log" 2>&1
python tools_dev/pmc_mfma.py "$(db $O/mfma)" adamw_kernel:3 "$O/pmc_mfma_util.json" > "$O/mfma_bench.txt"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d "$O/mfma_xs" -- python tools_dev/xslot_bench.py 256 300 3 49 3 3 > "$O/mfma_xs.log" 2>&1
python tools_dev/pmc_mfma.py "$(db $O/mfma_xs)" > "$O/mfma_xs.txt"
grep -h '"metric"' "$O/off.log" "$O/on.log" | tail -2 > "$O/bench_lines.json"
fi
if [[ " $PARTS " == *" config5 "* ]]; then
# BASELINE configs[4] (resnest50d, 100 x 3 slots, batch 256, bf16): kernel trace + the same counter passes
B5="python bench.py --config 5 --steps 3 --warmup 2 --no-cpu-baseline --no-prof"
SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d "$O/off5" -- python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-prof > "$O/off5.log" 2>&1
python tools_dev/rocpd_summary.py "$(db $O/off5)" adamw_kernel:3 > "$O/sum_off_config5.txt"
SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/fetch5" -- $B5 > "$O/fetch5.log" 2>&1
SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/write5" -- $B5 > "$O/write5.log" 2>&1
python tools_dev/pmc_traffic.py "$(db $O/fetch5)" "$(db $O/write5)" adamw_kernel:3 "$O/pmc_traffic_config5.json" 3 > "$O/pmc_traffic_config5.txt" 2> "$O/pmc_traffic_config5.err"
SCOUTER_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d "$O/mfma5" -- $B5 > "$O/mfma5.log" 2>&1
python tools_dev/pmc_mfma.py "$(db $O/mfma5)" adamw_kernel:3 "$O/pmc_mfma_util_config5.json" > "$O/mfma_bench_config5.txt"
fi
find "$O" -name "*.db" -delete; find "$O" -name "*.csv" -size +1M -delete
ls -la "$O"
