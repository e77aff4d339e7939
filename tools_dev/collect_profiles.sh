#!/bin/bash
# copies the summaries of a tools_dev/round_end.sh run (merged back under gpurun_out/) into profiles/<round>_*
# usage: bash tools_dev/collect_profiles.sh gpurun_out/r6p r06
O=$1; R=$2; P=$O/prof
cp 2>/dev/null $P/sum_off.txt profiles/${R}_kernel_trace_side_stream_off.txt
cp 2>/dev/null $P/sum_on.txt profiles/${R}_kernel_trace_side_stream_on.txt
cp 2>/dev/null $P/sum_off_config5.txt profiles/${R}_kernel_trace_config5_side_stream_off.txt
cp 2>/dev/null $P/pmc_traffic.json profiles/${R}_pmc_hbm_traffic.json; cp 2>/dev/null $P/pmc_traffic.txt profiles/${R}_pmc_hbm_traffic.txt
cp 2>/dev/null $P/pmc_traffic_config5.json profiles/${R}_pmc_hbm_traffic_config5.json; cp 2>/dev/null $P/pmc_traffic_config5.txt profiles/${R}_pmc_hbm_traffic_config5.txt
cp 2>/dev/null $P/pmc_mfma_util.json profiles/${R}_pmc_mfma_util.json; cp 2>/dev/null $P/mfma_bench.txt profiles/${R}_pmc_mfma_util_bench.txt
cp 2>/dev/null $P/pmc_mfma_util_config5.json profiles/${R}_pmc_mfma_util_config5.json; cp 2>/dev/null $P/mfma_bench_config5.txt profiles/${R}_pmc_mfma_util_bench_config5.txt
cp 2>/dev/null $P/mfma_xs.txt profiles/${R}_pmc_mfma_util_xslot.txt
cp 2>/dev/null $P/sum_xs_head.txt profiles/${R}_xslot_kernels_metric_head.txt
cp 2>/dev/null $P/sum_xs.txt profiles/${R}_xslot_kernels_batch256.txt; cp 2>/dev/null $P/sum_xs81.txt profiles/${R}_xslot_kernels_batch256_n81.txt
for c in 1 2 3 4 5; do cp 2>/dev/null $O/bench_config$c.json profiles/${R}_bench_config$c.json; done
cp 2>/dev/null $O/bench_config2_260.json profiles/${R}_bench_config2_260.json
cp 2>/dev/null $O/bench_line_1gpu.json profiles/${R}_bench_line_1gpu.json
ls profiles | grep ${R}_ | wc -l
