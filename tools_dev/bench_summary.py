"""usage: python tools_dev/bench_summary.py <bench-json-file>     (or: ... < file)"""
import json, sys
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("value %.1f %s  ms/step %.3f  n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
    print("roofline", d.get("roofline"))
    for k, v in d.get("kernels", {}).items():
        print("  %-30s %6.2f ms/step  %5.1f launches  avg %7.1f us  %s TF/s  %s GB/s(alg)" % (
            k, v["ms_per_step"], v["launches_per_step"], v["avg_us"], v["tflops"], v["gbps_algorithmic"]))
    if "cpu_baseline" in d:
        print("cpu_baseline", d["cpu_baseline"])
