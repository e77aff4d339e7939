# profiles + bench lines of the current tree (no pytest): bash tools_dev/r6_profiles.sh <outdir>
set -u
O=${1:-gpurun_out/r6p}
mkdir -p "$O"
bash tools_dev/refresh_profiles.sh "$O/prof" > "$O/refresh.log" 2>&1
mkdir -p profiles
R=r06
cp "$O/prof/pmc_traffic.json" profiles/${R}_pmc_hbm_traffic.json 2>/dev/null
cp "$O/prof/pmc_mfma_util.json" profiles/${R}_pmc_mfma_util.json 2>/dev/null
cp "$O/prof/pmc_traffic_config5.json" profiles/${R}_pmc_hbm_traffic_config5.json 2>/dev/null
cp "$O/prof/pmc_mfma_util_config5.json" profiles/${R}_pmc_mfma_util_config5.json 2>/dev/null
python bench.py --steps 40 2>/dev/null | tail -1 > "$O/bench_line_1gpu.json"
python bench.py --no-cpu-baseline --steps 40 --img-size 260 2>/dev/null | tail -1 > "$O/bench_config2_260.json"
for c in 1 3 4 5; do python bench.py --config $c --steps 20 2>/dev/null | tail -1 > "$O/bench_config$c.json"; done
cp "$O/bench_line_1gpu.json" "$O/bench_config2.json"
python tools_dev/bench_summary.py "$O/bench_line_1gpu.json" | head -3
for c in 1 3 4 5; do python -c "import json; d=json.load(open('$O/bench_config$c.json')); print($c, d['value'], d['ms_per_step'])"; done
