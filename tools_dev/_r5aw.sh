mkdir -p gpurun_out/r5aw
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5aw/e1.json 2> gpurun_out/r5aw/err.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --graph > gpurun_out/r5aw/g1.json 2>> gpurun_out/r5aw/err.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r5aw/e2.json 2>> gpurun_out/r5aw/err.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --graph > gpurun_out/r5aw/g2.json 2>> gpurun_out/r5aw/err.txt
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof --graph > gpurun_out/r5aw/g5.json 2>> gpurun_out/r5aw/err.txt
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/r5aw/e5.json 2>> gpurun_out/r5aw/err.txt
