mkdir -p gpurun_out/r5au
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5au/a.json 2> gpurun_out/r5au/err.txt
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5au/b.json 2>> gpurun_out/r5au/err.txt
python bench.py --config 5 --steps 20 > gpurun_out/r5au/c.json 2>> gpurun_out/r5au/err.txt
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5au/d.json 2>> gpurun_out/r5au/err.txt
