mkdir -p gpurun_out/r5am
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5am/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r5am/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5am/smoke.txt 2>&1
python bench.py > gpurun_out/r5am/bench_default.json 2> gpurun_out/r5am/bench_default.err
