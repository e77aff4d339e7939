mkdir -p gpurun_out/r6c
timeout 2000 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -v "Warning\|warn\|^$\|pin_memory" | tail -150 > gpurun_out/r6c/pytest_all.txt
