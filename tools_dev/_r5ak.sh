O=gpurun_out/r5_end6; mkdir -p $O
python bench.py --steps 40 2>/dev/null | tail -1 > "$O/bench_line_1gpu.json"
python bench.py --no-cpu-baseline --steps 40 --img-size 260 2>/dev/null | tail -1 > "$O/bench_config2_260.json"
for c in 1 3 4 5; do python bench.py --config $c --steps 20 2>/dev/null | tail -1 > "$O/bench_config$c.json"; done
cp "$O/bench_line_1gpu.json" "$O/bench_config2.json"
