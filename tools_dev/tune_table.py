"""Makes scouter_amd/tuning/gfx950.json, the STATIC block-tile / weight-gradient-plan table of the product path
(scouter_amd/kernels.py:_pick_tile).  Runs on an MI355X: one training step of every BASELINE config (and the 260x260
variants, the reference's --img_size default) with the timing autotuner on (SCOUTER_AUTOTUNE=1, 10 launches per candidate,
best of two batches); what the tuner chose is merged into the table.  The table is committed, so every process / rank /
profiler run launches the same kernel instance per layer shape.
usage: python tools_dev/tune_table.py [out.json]        (default: gpurun_out/gfx950.json; copy it to scouter_amd/tuning/)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gfx950.json"))
os.makedirs(os.path.dirname(out), exist_ok=True)
RUNS = [["--config", "2"], ["--config", "2", "--img-size", "260"], ["--config", "1"], ["--config", "1", "--img-size", "260"],
        ["--config", "4"], ["--config", "4", "--img-size", "260"], ["--config", "5", "--precision", "fp32"],
        ["--config", "5", "--precision", "bf16"], ["--config", "2", "--precision", "bf16"],
        ["--config", "2", "--batch", "35"], ["--config", "2", "--batch", "8"], ["--config", "1", "--batch", "8"]]
env = dict(os.environ, SCOUTER_AUTOTUNE="1", SCOUTER_TUNE_REPS="10", SCOUTER_TUNE_RECORD=out)
for r in RUNS:
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-prof", "--no-cpu-baseline"] + r
    rc = subprocess.call(cmd, env=env, stdout=subprocess.DEVNULL)
    print(" ".join(r), "->", rc, flush=True)
import json
print(len(json.load(open(out))["choices"]), "choices in", out)
