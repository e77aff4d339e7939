mkdir -p gpurun_out/r6k
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "Warning\|^$\|pin_memory\|Docs:" | tail -40 > gpurun_out/r6k/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/r6k/pytest_all.txt
