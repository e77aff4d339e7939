"""put a CLOCK under every "practical ceiling" sentence.

Runs training steps of the benchmark config, then -- in the same process, back to back -- the pure-MFMA burn kernels of
tools_dev/clock_probe.hip (fp32 32x32x2, then bf16 32x32x16) for >= 250 ms each and prints the shader clock the chip
sustains under that load (s_memtime cycles / s_memrealtime ticks, sampled inside the kernel) with the matrix
throughput actually delivered, against the nominal 2.4 GHz the roofline peaks assume.

usage (GPU box): python tools_dev/clocks.py [ms per burn] > gpurun_out/clocks.txt"""
import ctypes
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn   # noqa: E402

MS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
so = os.path.join(ROOT, "tools_dev", "clock_probe.bin")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(so[:-4] + ".hip"):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", so[:-4] + ".hip", "-o", so])
import __graft_entry__ as G   # noqa: E402
G.build()
from scouter_amd.optim import FusedAdamW   # noqa: E402
from scouter_amd.sloter.slot_model import SlotModel   # noqa: E402
probe = ctypes.CDLL(so)
dev = torch.device("cuda:0")
cfg = dict(Bn.CONFIGS[2])
torch.manual_seed(0)
model = SlotModel(Bn.make_args(cfg)).to(dev).train()
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4)
x, y = Bn.synth_batch(cfg["batch"], cfg["img_size"], cfg["num_classes"], 0, dev, 3)


def steps(n):
    for _ in range(n):
        opt.zero_grad()
        out, losses = model(x, y)
        losses[0].backward()
        opt.step()


BLOCKS, MAXS = 512, 16384
samples = torch.zeros(3 * MAXS + 1, dtype=torch.int64, device=dev)
per_wave = torch.zeros(BLOCKS * 4, dtype=torch.int64, device=dev)
sink = torch.zeros(BLOCKS * 256, dtype=torch.float32, device=dev)


def burn(kind, ms):
    samples.zero_(); per_wave.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = probe.clock_probe_launch(kind, ms, BLOCKS, ctypes.c_void_p(samples.data_ptr()), MAXS,
                                  ctypes.c_void_p(per_wave.data_ptr()), ctypes.c_void_p(sink.data_ptr()), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    host_s = time.perf_counter() - t0
    s = samples.cpu().numpy()
    n = int(s[3 * MAXS])
    rt, cyc = s[0:3 * n:3].astype(float), s[1:3 * n:3].astype(float)
    ticks_per_s = rt[-1] / host_s              # s_memrealtime rate, calibrated against the host clock (nominal 100 MHz)
    total = float(per_wave.sum())
    flop_per = 4096.0 if kind in (0, 2) else 32768.0
    rate = ticks_per_s if abs(ticks_per_s / 1e8 - 1.0) > 0.02 else 1e8     # (trust the nominal 100 MHz if the host agrees)
    tf = total * flop_per / (rt[-1] / rate) / 1e12
    out = []
    k = max(1, n // 12)
    for i in range(k, n, k):
        ghz = (cyc[i] - cyc[i - k]) / ((rt[i] - rt[i - k]) / rate) / 1e9
        out.append("%.0f ms: %.3f GHz" % (rt[i] / rate * 1e3, ghz))
    mean_ghz = (cyc[-1] - cyc[0]) / ((rt[-1] - rt[0]) / rate) / 1e9
    return dict(host_ms=host_s * 1e3, ticks_per_s=ticks_per_s, tflops=tf, mean_ghz=mean_ghz, trace=out)


steps(3)
torch.cuda.synchronize()
print("# tools_dev/clocks.py: sustained shader clock under pure MFMA load, right after training steps in the same process")
print("# device:", torch.cuda.get_device_name(0), "| burn", MS, "ms per kernel,", BLOCKS, "workgroups x 4 waves (2 per SIMD)")
for kind, name, peak in ((0, "fp32 v_mfma_f32_32x32x2_f32, CONSTANT operands", 157.3),
                         (2, "fp32 v_mfma_f32_32x32x2_f32, RANDOM operands changing per MFMA", 157.3),
                         (1, "bf16 v_mfma_f32_32x32x16_bf16, CONSTANT operands", 2500.0),
                         (3, "bf16 v_mfma_f32_32x32x16_bf16, RANDOM operands changing per MFMA", 2500.0)):
    for rep in range(2):
        steps(10)                           # the training step itself as the warm-up / thermal state
        r = burn(kind, MS)
        print("%s run %d: mean %.3f GHz over %.0f ms (s_memrealtime calibrates to %.2f MHz vs host), delivered %.1f TFLOP/s "
              "= %.3f of the %.1f peak; at 2.4 GHz nominal this load would deliver %.1f"
              % (name, rep, r["mean_ghz"], r["host_ms"], r["ticks_per_s"] / 1e6, r["tflops"], r["tflops"] / peak, peak,
                 r["tflops"] * 2.4 / r["mean_ghz"]))
        print("   clock trace: " + " | ".join(r["trace"]))
# and the clock while the training step itself runs: cycles of a resident observer are not available from outside a
# kernel, so take rocm-smi's view during a burst of steps
try:
    import threading
    seen = []

    def poll():
        for _ in range(12):
            o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout
            for ln in o.splitlines():
                if "sclk" in ln:
                    seen.append(ln.strip())
                    break
            time.sleep(0.05)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    while th.is_alive():
        steps(5)
    torch.cuda.synchronize()
    th.join()
    print("rocm-smi sclk samples while training steps run:")
    for ln in seen:
        print("   " + ln)
except Exception as e:   # noqa: BLE001
    print("rocm-smi polling failed:", e)
