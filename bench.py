#!/usr/bin/env python
"""Headline benchmark: images/sec of the full xSlot training step (BASELINE.json configs[1]: resnest26d + positive
xSlot, 10 classes, channel 2048, to_k_layer 3, power 2, T=3, 224x224, per-GPU batch 70, fp32) on N MI355X.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W            (one rank per GPU, gradients all-reduced over RCCL/xGMI)
  python bench.py --config {1,2,3,4,5} [--precision fp32|bf16]   the other BASELINE configs at their per-GPU batch
         (1-based, SURVEY.md section 8 table: 1 = MNIST resnet18, 2 = the headline, 3 = negative xSlot, 4 = CUB200,
         5 = ImageNet-100 resnest50d 100x3 slots -- the config that names bf16, its default precision); same JSON schema

A step = zero_grad -> forward -> loss -> backward (+ gradient all-reduce) -> AdamW, on a synthetic batch already
resident in HBM.  Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      the dominant kernel class (fp32-MFMA implicit-GEMM convolution): algorithmic FLOPs / hipEvent time
                measured over the timed region on the launch stream, vs the 157.3 TFLOP/s fp32 matrix peak
  cpu_baseline  the CPU oracle (oracle/torch_oracle.py, a port of the reference's PyTorch path) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N=1 only)."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(model="resnest26d", num_classes=10, slots_per_class=1, channel=2048, to_k_layer=3, power=2, loss_status=1,
           lambda_value="1", hidden_dim=64, img_size=224, batch=70, dataset="ImageNet", precision="fp32",
           fwd_gflop=7.243 + 0.0159,
           name="BASELINE configs[1]: ImageNet-10 resnest26d + positive xSlot, channel 2048, T=3, to_k_layer 3, power 2",
           metric="images/sec training step (resnest26d+xSlot, 224^2, bs70)")
# BASELINE.json configs, 1-based as in SURVEY.md section 8 (head recipes from the reference README, FLOPs per image
# = backbone conv + head forward); --config 2 is the headline (the default)
CONFIGS = {
    1: dict(CFG, model="resnet18", channel=512, to_k_layer=1, power=1, batch=64, dataset="MNIST",
            fwd_gflop=3.406 + 0.0054,
            name="BASELINE configs[0]: MNIST-shaped resnet18 (1-channel 3x3 stem) + xSlot, 10 classes, T=3, to_k_layer 1, power 1",
            metric="images/sec training step (resnet18+xSlot MNIST stem, 224^2, bs64)"),
    2: CFG,
    3: dict(CFG, loss_status=-1,
            name="BASELINE configs[2]: ImageNet-10 resnest26d + NEGATIVE xSlot (loss_status=-1), channel 2048, T=3, to_k_layer 3, power 2",
            metric="images/sec training step (resnest26d+negative xSlot, 224^2, bs70)"),
    4: dict(CFG, num_classes=200, batch=128, fwd_gflop=7.243 + 0.0511,
            name="BASELINE configs[3]: CUB200 resnest26d + xSlot, 200 classes x 1 slot, channel 2048, T=3, to_k_layer 3, power 2",
            metric="images/sec training step (resnest26d+xSlot 200 slots, 224^2, bs128)"),
    5: dict(CFG, model="resnest50d", num_classes=100, slots_per_class=3, batch=256, precision="bf16",
            fwd_gflop=10.738 + 0.0696,
            name="BASELINE configs[4]: ImageNet-100 resnest50d + xSlot, 100 classes x 3 slots (wide-slot stress), channel 2048, T=3, to_k_layer 3, power 2",
            metric="images/sec training step (resnest50d+xSlot 300 slots, 224^2, bs256)"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md chip table (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0              # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), same table


# forward convolution GFLOP per image of the backbones (probed on the reference's timm
# models), by (model, image size)
BACKBONE_GFLOP = {("resnest26d", 224): 7.243, ("resnest26d", 260): 10.302, ("resnest50d", 224): 10.738,
                  ("resnest50d", 260): 15.403, ("resnet18", 224): 3.406, ("resnet18", 260): 4.978}


def fwd_gflop(cfg):
    """Algorithmic forward GFLOP per image: backbone convolutions + the head (conv1x1 channel -> d, to_k MLP, T x (QK^T +
    AV), (T-1) GRUs)."""
    n = (-(-cfg["img_size"] // 32)) ** 2
    d, S, T = cfg["hidden_dim"], cfg["num_classes"] * cfg["slots_per_class"], 3
    head = 2.0 * n * cfg["channel"] * d + 2.0 * cfg["to_k_layer"] * n * d * d + T * 4.0 * S * n * d + (T - 1) * 12.0 * S * d * d
    return BACKBONE_GFLOP[(cfg["model"], cfg["img_size"])] + head * 1e-9


def make_args(cfg):
    return argparse.Namespace(model=cfg["model"], pre_trained=False, num_classes=cfg["num_classes"],
                              dataset=cfg.get("dataset", "ImageNet"),
                              use_slot=True, use_pre=False, grad=False, channel=cfg["channel"],
                              slots_per_class=cfg["slots_per_class"], hidden_dim=cfg["hidden_dim"], freeze_layers=0,
                              vis=False, vis_id=0, loss_status=cfg["loss_status"], power=cfg["power"],
                              to_k_layer=cfg["to_k_layer"], lambda_value=cfg["lambda_value"],
                              precision=cfg.get("precision", "fp32"))


def synth_batch(B, size, C, rank, device, in_chans=3):
    rng = np.random.default_rng(1234 + rank)
    x = torch.from_numpy(rng.standard_normal((B, in_chans, size, size), dtype=np.float32)).to(device)
    y = torch.from_numpy(rng.integers(0, C, B).astype(np.int64)).to(device)
    return x, y


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, budget_s=20.0, steps=3):
    """The CPU oracle's train step (the port of the reference's PyTorch-CPU path) beside the GPU number: SURVEY.md
    section 8d asks for the config's own batch, 1 warm-up + 3 timed steps, all host cores, CPU model and core count in
    the object.  The sample is bounded to ~`budget_s` of CPU work: a probe step at batch 8 sizes the sample batch (the
    config batch when 4 steps of it fit the budget, else the largest batch that does -- said in `sample`)."""
    from oracle import torch_oracle as O
    ncpu = os.cpu_count() or 1
    mnist = cfg.get("dataset") == "MNIST"
    spec = O.state_dict_spec(cfg["model"], cfg["num_classes"], cfg["slots_per_class"], cfg["to_k_layer"],
                             in_chans=1 if mnist else 3, mnist_stem=mnist)
    P = O.synth_state(spec, 0)
    ocfg = dict(model=cfg["model"], num_classes=cfg["num_classes"], slots_per_class=cfg["slots_per_class"],
                loss_status=cfg["loss_status"], power=cfg["power"], lambda_value=float(cfg["lambda_value"]))
    tr = O.OracleTrainer(P, ocfg, lr=1e-4)

    def timed(b, n):
        img, lab = O.synth_batch(b, 1 if mnist else 3, cfg["img_size"], cfg["num_classes"], 1234)
        t0 = time.time()
        for _ in range(n):
            tr.step(img, lab)
        return (time.time() - t0) / n
    # thread count: 32.  SURVEY 8d asks for all host cores, but on the GPU box (2 x EPYC 9575F, 256 logical CPUs) torch-CPU
    # convolutions at these batch sizes are SLOWER beyond a few dozen threads: the round-2 probe timed {32, 64, 256}
    # threads on configs 1-4 and 32 won every time (256 threads: minutes per step) -- the CPU gets its best setting.
    cores = min(32, ncpu)
    torch.set_num_threads(cores)
    timed(8, 1)                                          # thread pool / allocator warm-up
    probe = timed(8, 1) / 8                              # seconds per image
    full = cfg["batch"]
    sample_b = full if probe * full * (steps + 1) <= budget_s else max(8, int(budget_s / (probe * (steps + 1))) // 8 * 8)
    sample_b = min(sample_b, full)
    timed(sample_b, 1)                                   # the warm-up step at the sample batch
    dt = timed(sample_b, steps)
    why = "the config's own batch" if sample_b == full else \
        "batch %d instead of the config's %d to keep the default run within ~%d s of CPU work" % (sample_b, full, budget_s)
    return {"value": round(sample_b / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_name(), "host_logical_cpus": ncpu, "seconds_per_step": round(dt, 3),
            "sample": "%d timed train steps (+1 warm-up) of the torch-CPU oracle (oracle/torch_oracle.py, fp32), batch %d "
                      "of the same %s %dx%d workload (%s), torch.set_num_threads(%d) of %d logical CPUs (32 measured faster than 64 / all) "
                      "on %s" % (steps, sample_b, cfg["model"], cfg["img_size"], cfg["img_size"], why, cores, ncpu,
                                 cpu_model_name())}


def xslot_roofline(device, batch=256, slots=300, spc=3, tokens=49, iters=3, layers=3):
    """north_star target: the fused xSlot forward at batch 256 (BASELINE configs[4]'s head: 100 classes x 3 slots, 7x7
    grid, to_k_layer 3) against the fp32 MFMA roofline.  Kernel time = the library's hipEvents around each launch, median of 8
    batches of 20 launches after 200 warm-up launches (the clock needs ~150 launches of THIS kernel to settle after the
    training steps: 135 -> 122 us); FLOPs are algorithmic (no padding counted)."""
    from scouter_amd import _native, kernels as K
    g = torch.Generator(device=device).manual_seed(0)
    r = lambda *sh: torch.randn(*sh, device=device, generator=g)
    d = 64
    X, PE = r(batch, tokens, d).relu_(), r(tokens, d) * 0.3
    tw, tb = [r(d, d) * 0.1 for _ in range(layers)], [r(d) * 0.1 for _ in range(layers)]
    s0 = r(slots, d).abs() * 0.5
    wih, whh, bih, bhh = r(3 * d, d) * 0.1, r(3 * d, d) * 0.1, r(3 * d) * 0.1, r(3 * d) * 0.1
    fn = lambda: K.xslot_fwd(X, PE, tw, tb, s0, wih, whh, bih, bhh, spc, iters, 1)
    L = _native.lib()
    buf = ctypes.create_string_buffer(1 << 14)

    def median_launch(fn, kernel):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        L.scouter_prof_collect(buf, len(buf))
        times = []
        for _ in range(8):
            L.scouter_prof_enable(1)
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            L.scouter_prof_enable(0)
            L.scouter_prof_collect(buf, len(buf))
            for row in buf.value.decode().splitlines():
                name, n, ms = row.split("\t")[:3]
                if name == kernel:
                    times.append(float(ms) / float(n) * 1e-3)
        return sorted(times)[len(times) // 2]

    t = median_launch(fn, "xslot_fwd")
    qk = 2.0 * slots * tokens * d
    fl = batch * (2.0 * layers * tokens * d * d + iters * 2 * qk + (iters - 1) * 12.0 * slots * d * d)
    saved = fn()
    dlog, g_area = r(batch, slots // spc), torch.full((1,), 1e-3, device=device)
    tb_ = median_launch(lambda: K.xslot_bwd(X, PE, tw, s0, wih, whh, bih, bhh, saved, dlog, g_area, spc, iters, 1),
                        "xslot_bwd")
    backward = {"kernel": "xslot_bwd_kernel (per-iteration forward recomputation + GRU / normaliser / attention backward + "
                          "slot-index contractions + to_k MLP backward, fused)", "avg_launch_us": round(tb_ * 1e6, 1),
                "flop_model": "3 x forward (recomputation + two GEMMs per forward GEMM)",
                "achieved": round(3 * fl / tb_ / 1e12, 2), "frac": round(3 * fl / tb_ / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
    fwd_kernel = "xslot_fwd_kernel (to_k MLP + 3 x [QK^T, sigmoid, AV, GRU] fused, v_mfma_f32_32x32x2_f32)"
    if slots <= 16 and os.environ.get("SCOUTER_XSLOT_SMALL", "1") != "0":
        # heads with <= 16 slots per image (the metric's own: 10 classes x 1 slot) run the small-S instantiation: a latency
        # chain per image (one workgroup, four waves), not a throughput kernel -- the launch time is the figure, the
        # roofline fraction is reported for completeness
        fwd_kernel = ("xslot_small_fwd_kernel (same fused forward on v_mfma_f32_16x16x4_f32 tiles: slots on the 16 MFMA columns, "
                      "token tiles / GRU hidden units split over four waves, GRU weights resident in registers)")
        backward["kernel"] = ("xslot_small_bwd_kernel (same fused backward, 16x16x4 tiles, both GRU weight orientations in "
                              "registers, four LDS hand-offs per iteration)")
    return {"backward": backward, "kernel": fwd_kernel,
            "batch": batch, "slots": slots, "tokens": tokens, "to_k_layers": layers, "avg_launch_us": round(t * 1e6, 1),
            "achieved": round(fl / t / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(fl / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "bound": "mfma",
            "measured": "hipEvents around each launch, median of 8 batches x 20 launches after 200 warm-up launches"}


# `sustained` (TFLOP/s, reported BESIDE `peak`, never instead of it): what a pure-MFMA burn with RANDOM operands that change
# from MFMA to MFMA delivers on this chip for >= 300 ms right after training steps (tools_dev/clocks.py,
# profiles/r04_clocks.txt): the fp32 MFMA holds 2.40 GHz (154 TFLOP/s = 0.98 of nominal), the bf16 MFMA is power-managed down
# to 1.88 GHz with real data (1 886 TFLOP/s = 0.754 of the 2 500 nominal; 2.39 GHz / 2 497 only with constant operands), and
# the plane kernels themselves run at 1.62-1.88 GHz (cycle stamps inside pwgrad_taps_kernel, tools_dev/pwt_stamps.py).
# committed PMC passes (tools_dev/refresh_profiles.sh) per BASELINE config at its own batch / image size / precision
PMC_FILES = {2: ("r06_pmc_hbm_traffic.json", "r06_pmc_mfma_util.json"),
             5: ("r06_pmc_hbm_traffic_config5.json", "r06_pmc_mfma_util_config5.json")}
HBM_PEAK_TBS, HBM_ACHIEVABLE_TBS = 8.0, 6.3          # MI355X_MICROARCH.md: HBM3E nominal / what a streaming kernel reaches
# Kernel classes of the roofline object: bench label prefixes (the library's hipEvent scopes) and the rocprofv3 kernel-name
# prefixes of the SAME kernels.  Names are matched by exact prefix ("void wgrad_kernel<" does not match
# "void pwgrad_kernel<": round 2's substring match mixed the two).
CLASSES = {
    "fp32_mfma_conv": {
        "labels": ("igemm_fwd<", "igemm_dgrad<", "igemm_dgrad+bn_bwd<", "wgrad<", "wgrad_taps"),
        "rocprof": ("void igemm_kernel<", "void wgrad_kernel<", "void wgrad_taps_kernel<", "void pwp_kernel<",
                    "void pwp_fused_kernel<"),
        "peak": 157.3, "sustained": 154.0, "what": "fp32 implicit-GEMM convolutions (forward, input gradient incl. the fused BatchNorm-backward "
                               "epilogue, weight gradient), v_mfma_f32_32x32x2_f32"},
    "bf16x3_plane_conv": {
        "labels": ("pconv_fwd<bf16x3>", "pconv_dgrad<bf16x3>", "pwgrad<bf16x3>", "xconv_fwd<bf16x3>", "xconv_dgrad<bf16x3>",
                   "xconv_dgrad+bn_bwd<bf16x3>", "xwgrad<bf16x3>", "xconv3_fwd<bf16x3>", "xconv3_dgrad<bf16x3>",
                   "xconv3_dgrad+bn_bwd<bf16x3>", "xpw_dgrad+bn_bwd<bf16x3>", "xpw_fwd<bf16x3>", "xhalo_fwd<bf16x3>",
                   "xhalo_dgrad<bf16x3>", "xhalo_dgrad+bn_bwd<bf16x3>", "xwgrad_taps<bf16x3>"),
        "rocprof": ("void pconv_kernel<", "void phalo_kernel<", "void pwgrad_kernel<", "void ppersist_kernel<",
                    "void pwgrad_taps_kernel<", "void xgemm_kernel<", "void xwgrad_kernel<", "void xpw_fused_kernel<", "void xpw_fwd_kernel<",
                    "void xhalo_kernel<"),
        "peak": 2500.0 / 6.0, "sustained": 1886.0 / 6.0, "what": "grouped 3x3 convolutions (pre-split operand planes) and, round 5, the deep 1x1 "
                                      "convolutions (activation split in registers) on exact three-way bf16 operand splits: 6 x "
                                      "v_mfma_f32_32x32x16_bf16 per fp32-grade product, fp32 accumulate; peak = 2500 / 6 "
                                      "TFLOP/s of algorithmic work"},
    "bf16_mfma_conv": {
        "labels": ("igemm_fwd_bf16<", "igemm_dgrad_bf16<", "igemm_dgrad_bf16+bn_bwd<", "wgrad_bf16", "pconv_fwd<bf16>",
                   "pconv_dgrad<bf16>", "pwgrad<bf16>", "bwgrad_taps<bf16>", "bhalo_dgrad<bf16>", "bhalo_fwd<bf16>"),
        "rocprof": ("void igemm_bf16_kernel<", "void wgrad_bf16_kernel<", "void bwgrad_taps_kernel<", "bhalo_dgrad_kernel", "void bhalo_fwd_kernel<", "void pconv_kernel<", "void phalo_kernel<",
                    "void pwgrad_kernel<", "void pwgrad_taps_kernel<", "void ppersist_kernel<", "void pwb_fused_kernel<",
                    "void pwb_fwd_kernel<"),
        "peak": 2500.0, "sustained": 1886.0, "what": "bf16-input convolutions (--precision bf16), v_mfma_f32_32x32x16_bf16, fp32 accumulate"},
}


# The PERSISTENT pointwise kernels (weights resident in LDS, the activation streamed through an LDS ring, register epilogue) serve
# the short-K 1x1 layers: 6-18 bytes per output element against 2 K FLOP -- streams with a small GEMM attached, bound by HBM
# whatever matrix instruction multiplies.  They are a class of their own, priced against the HBM roofline (algorithmic bytes /
# time vs 8 TB/s nominal, 6.3 TB/s a streaming kernel reaches), and are NOT counted in the matrix classes above.
HBM_CLASS = {
    "labels": ("igemm_fwd<persistent>", "igemm_dgrad<persistent>", "igemm_dgrad+bn_bwd<persistent>", "xpw_dgrad+bn_bwd<bf16x3>",
               "xpw_fwd<bf16x3>", "igemm_fwd_bf16<persistent>", "igemm_dgrad_bf16<persistent>"),
    "rocprof": ("void pwp_kernel<", "void pwp_fused_kernel<", "void xpw_fused_kernel<", "void xpw_fwd_kernel<",
                "void pwb_fused_kernel<", "void pwb_fwd_kernel<", "void pwb_dgrad_kernel<"),
    "what": "persistent pointwise kernels of the short-K 1x1 layers (forward with fused statistics, input gradient with the fused "
            "BatchNorm-backward epilogue): HBM streams with a small GEMM attached (fp32 MFMA / bf16x3 / bf16 MFMA)"}


def _profile_json(fname):
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        doc = json.load(f)
    # the PMC passes describe the kernel instances of the tile table they ran with: a profile taken under another table is
    # stale -- dropped, the line then carries traffic / pmc_mfma_busy = null rather than wrong numbers
    import hashlib
    tp = os.path.join(ROOT, "scouter_amd", "tuning", "gfx950.json")
    sha = hashlib.sha256(open(tp, "rb").read()).hexdigest()[:16] if os.path.exists(tp) else None
    if doc.get("tuning_sha16") != sha:
        return None
    return doc


def pmc_class(prefixes, files):
    """(HBM-side bytes per launch, MFMA-busy fraction, sources) of the kernels whose rocprofv3 names start with one of
    `prefixes`, from the committed PMC passes over THIS command (profiles/r05_pmc_*.json, made by tools_dev/pmc_traffic.py
    -- 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md -- and tools_dev/pmc_mfma.py); PMC
    counters cannot be collected from inside the timed process.  The static tile table makes the instances of those
    runs the instances of this one.  None where a file is missing."""
    traffic = busy = None
    src = []
    t = _profile_json(files[0])
    if t:
        rows = [r for n, r in t["kernels"].items() if n.startswith(prefixes)]
        nl = sum(r["launches"] for r in rows)
        if nl:
            traffic = round(sum((r["read_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches"] for r in rows) / nl)
            src.append("profiles/" + files[0])
    m = _profile_json(files[1])
    if m:
        rows = [r for n, r in m["kernels"].items() if n.startswith(prefixes)]
        tot = sum(r["avg_us"] * r["launches"] for r in rows)
        if tot:
            busy = round(sum(r["mfma_busy"] * r["avg_us"] * r["launches"] for r in rows) / tot, 4)
            src.append("profiles/" + files[1])
    return traffic, busy, src


def hbm_step_roofline(files, ms_per_step):
    """HBM-roofline fraction of the WHOLE step: counter bytes of every kernel of a step (the committed FETCH_SIZE / WRITE_SIZE
    passes of this command) over the step time measured here, against the nominal 8 TB/s and the 6.3 TB/s a streaming kernel
    reaches -- the figure that matters for BASELINE configs[4], whose step is bound by its elementwise passes."""
    t = _profile_json(files[0])
    if not t or not t.get("hbm_bytes_per_step"):
        return None
    tbs = t["hbm_bytes_per_step"] / (ms_per_step * 1e-3) / 1e12
    return {"bound": "hbm", "bytes_per_step": round(t["hbm_bytes_per_step"]), "achieved": round(tbs, 3), "unit": "TB/s",
            "peak": HBM_PEAK_TBS, "frac": round(tbs / HBM_PEAK_TBS, 4), "achievable": HBM_ACHIEVABLE_TBS,
            "frac_of_achievable": round(tbs / HBM_ACHIEVABLE_TBS, 4), "source": "profiles/" + files[0],
            "note": "HBM-side counter bytes of all kernels of one step / this run's ms_per_step (kernels overlap on streams)"}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU, RCCL) the way the reference's README
    does with torch.distributed.launch (reference README.md:20-22, tools/prepare_things.py:9-31) and pass their single
    JSON line through.  Refuses when the node has fewer than N devices."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit("bench.py --gpus %d: this node exposes %d HIP device(s); refusing to run fewer ranks than "
                         "asked for (the reported n_gpus would be wrong)" % (n, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --standalone: torchrun's own c10d rendezvous binds a free port and KEEPS it (no bind-close-reuse race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + argv
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE config, 1-based (SURVEY.md section 8 table); 2 = the headline metric")
    ap.add_argument("--precision", choices=("fp32", "bf16"), default=None,
                    help="backbone matrix-input precision (default: fp32; bf16 for --config 5, the config that names it)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's; weak scaling)")
    ap.add_argument("--img-size", type=int, default=None, choices=(224, 260),
                    help="input resolution (default 224, the metric's; 260 = the reference's --img_size default, 9x9 tokens)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured hipGraph (scouter_amd/graph.py) instead of issuing it kernel by "
                         "kernel; measured SLOWER on ROCm 7.0 / MI355X for this GPU-bound step, hence opt-in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-kernel hipEvent pass (no roofline object)")
    ap.add_argument("--prof-steps", type=int, default=5, help="steps of the serial per-kernel timing pass")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a.gpus, sys.argv[1:])
    import __graft_entry__ as G
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and not os.environ.get("SCOUTER_FORCE_DP"):
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    force_dp = bool(os.environ.get("SCOUTER_FORCE_DP"))      # dev: exercise the RCCL path with a 1-rank group
    if world > 1 or (force_dp and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)
    # build (or verify) the HIP library on rank 0 only, then let the other ranks load the finished file
    if rank == 0:
        G.build()
    if dist.is_initialized():
        dist.barrier()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    from scouter_amd import _native
    from scouter_amd.optim import FusedAdamW
    from scouter_amd.parallel import DistributedDataParallel
    from scouter_amd.sloter.slot_model import SlotModel

    cfg = dict(CONFIGS[a.config])
    if a.batch:
        cfg["batch"] = a.batch
    if a.precision:
        cfg["precision"] = a.precision
    if a.img_size:
        cfg["img_size"] = a.img_size
    cfg["fwd_gflop"] = fwd_gflop(cfg)
    bf16 = cfg["precision"] == "bf16"
    mnist = cfg["dataset"] == "MNIST"
    torch.manual_seed(0)
    model = SlotModel(make_args(cfg))
    for m in model.modules():                          # exercise every layer: last-BN gamma = 1
        if hasattr(m, "zero_init_last_bn"):
            torch.nn.init.ones_((m.bn3 if hasattr(m, "bn3") else m.bn2).weight)
    model = model.to(device).train()
    net = DistributedDataParallel(model, device_ids=[local_rank]) if (world > 1 or dist.is_initialized()) else model
    # --graph (one GPU): the whole step is replayed from a captured hipGraph (bit-identical to the eager step,
    # tests/test_graph_gpu.py).  N > 1 stays eager -- capturing the RCCL collectives has not been validated on a
    # multi-GPU node from here (SCOUTER_GRAPH_DP=1 opts in).
    use_graph = a.graph and (world == 1 and not dist.is_initialized() or bool(os.environ.get("SCOUTER_GRAPH_DP")))
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, capturable=use_graph)
    x, y = synth_batch(cfg["batch"], cfg["img_size"], cfg["num_classes"], rank, device, 1 if mnist else 3)
    graphed = None
    if use_graph:
        from scouter_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(net, opt)

    def step(eager=False):
        if graphed is not None and not eager:
            logp, stats = graphed(x, y)
            return [stats[0]]
        opt.zero_grad()
        out, losses = net(x, y)
        losses[0].backward(model.loss_seed(losses[0]))      # (as engine.calculation: cached ones instead of autograd's fill)
        opt.step()
        return losses

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    # development (tools_dev/priority_ab.sh): the whole step on a high-priority compute stream instead of the default stream
    if os.environ.get("SCOUTER_MAIN_PRIORITY"):
        main_stream = torch.cuda.Stream(device=device, priority=int(os.environ["SCOUTER_MAIN_PRIORITY"]))
        main_stream.wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(main_stream)

    # setup, outside the W warm-up steps: the first pass through every layer shape picks its block tile (each candidate is
    # timed once, scouter_amd/kernels.py:_pick_tile) -- like building the model, it happens once per process
    step()
    fence()
    for _ in range(a.warmup):
        step()
    L = _native.lib()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    t_host = time.perf_counter() - t0           # the host has ENQUEUED the K steps (diagnostic: close to dt = launch-bound)
    fence()
    dt = time.perf_counter() - t0
    # Per-kernel timing pass (same process, model, batch): hipEvents around every launch, on its stream.  It runs with
    # the weight-gradient side stream DISABLED: in the timed region above wgrad kernels overlap BatchNorm-backward /
    # dgrad kernels on purpose, which inflates each kernel's elapsed time and would misstate the kernels' own quality.
    prof_buf = ctypes.create_string_buffer(1 << 16)
    prof_steps = 0 if a.no_prof else max(1, a.prof_steps)
    if prof_steps:
        model.set_side_stream(False)
        step(eager=True)
        fence()
        L.scouter_prof_enable(1)
        for _ in range(prof_steps):
            step(eager=True)
        fence()
        L.scouter_prof_enable(0)
        L.scouter_prof_collect(prof_buf, len(prof_buf))
        model.set_side_stream(True)
    loss_val = float(losses[0].detach())
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dp_info = None
    if dist.is_initialized():
        # what explains a scaling curve: the group size RCCL actually ran with, every rank's own time, and
        # the time the compute stream spent WAITING for the gradient all-reduces (hipEvents around the waits, a few extra steps)
        net.measure_exposed = True
        for _ in range(5):
            step(eager=True)
        fence()
        exposed = net.exposed_allreduce_ms()
        net.measure_exposed = False
        mine = torch.tensor([dt, exposed if exposed is not None else -1.0, t_host], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ms = [1e3 * float(t[0]) / a.steps for t in every]
        ex = [float(t[1]) for t in every]
        dp_info = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                   "ms_per_step_by_rank": [round(v, 3) for v in ms], "ms_per_step_spread": round(max(ms) - min(ms), 3),
                   "allreduce_exposed_ms": round(max(ex), 4), "allreduce_exposed_ms_by_rank": [round(v, 4) for v in ex],
                   # wall time until each rank's host had ENQUEUED its K timed steps, per step: a rank whose figure is close to
                   # its ms_per_step is launch-bound (N processes share the host's cores)
                   "host_enqueue_ms_per_step_by_rank": [round(1e3 * float(t[2]) / a.steps, 3) for t in every],
                   "gradient_bytes_per_step": int(model.grad_arena().numel * 4),
                   "buckets": int(net.buckets_last_step),
                   "allreduce_exposed_measured": "hipEvents around the compute stream's wait for the gradient collectives on 5 "
                                                 "extra EAGER steps after the timed region (also under --graph)",
                   "buffer_broadcast": "asynchronous in every forward, consumed in front of the first BatchNorm"}
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    if rank == 0:
        imgs = world * cfg["batch"] * a.steps
        value = imgs / dt
        kern = {}
        for row in prof_buf.value.decode().splitlines():
            name, n, ms, fl, by = row.split("\t")
            n, ms, fl, by = float(n), float(ms), float(fl), float(by)
            kern[name] = {"launches_per_step": n / prof_steps, "ms_per_step": round(ms / prof_steps, 4),
                          "avg_us": round(1e3 * ms / n, 2), "tflops": round(fl / (ms * 1e-3) / 1e12, 2) if fl else None,
                          "gbps_algorithmic": round(by / (ms * 1e-3) / 1e9, 1) if by else None}
        # ---- roofline: stable kernel CLASSES (not an argmax over near-tied instances): each with its algorithmic FLOPs,
        # time, fraction of ITS matrix peak and -- from the committed PMC passes of this command -- MFMA-busy and HBM-side
        # traffic per launch against the algorithmic bytes.  `roofline` itself = the class with the most time per step.
        # (the committed PMC passes exist for configs 2 and 5 at their own batch / image size / precision)
        pmc_files = PMC_FILES.get(a.config) if (cfg["img_size"] == 224 and cfg["batch"] == CONFIGS[a.config]["batch"] and
                                                cfg["precision"] == CONFIGS[a.config].get("precision", "fp32")) else None
        classes = {}
        for cname, c in CLASSES.items():
            rows = [k for k in kern if k.startswith(c["labels"]) and not k.startswith(HBM_CLASS["labels"]) and kern[k]["tflops"]]
            if not rows:
                continue
            ms = sum(kern[k]["ms_per_step"] for k in rows)
            fl = sum(kern[k]["tflops"] * kern[k]["ms_per_step"] for k in rows)            # GFLOP per step
            nl = sum(kern[k]["launches_per_step"] for k in rows)
            by = sum((kern[k]["gbps_algorithmic"] or 0.0) * kern[k]["ms_per_step"] * 1e6 for k in rows)   # bytes per step
            traffic, busy, src = pmc_class(c["rocprof"], pmc_files) if pmc_files else (None, None, [])
            classes[cname] = {"kernels": c["what"], "instances": sorted(rows), "gflop_per_step": round(fl, 1),
                              "ms_per_step": round(ms, 4), "launches_per_step": nl, "avg_launch_us": round(1e3 * ms / nl, 2),
                              "achieved": round(fl / ms, 2), "peak": round(c["peak"], 1), "unit": "TFLOP/s",
                              "frac": round(fl / ms / c["peak"], 4),
                              "peak_at_sustained_clock": round(c["sustained"], 1),
                              "frac_of_sustained": round(fl / ms / c["sustained"], 4), "pmc_mfma_busy": busy,
                              "algorithmic_bytes_per_launch": round(by / nl), "traffic": traffic,
                              "traffic_over_algorithmic": round(traffic / (by / nl), 2) if traffic and by else None,
                              "pmc_source": src or None}
        # the HBM-bound stream class (see HBM_CLASS): algorithmic bytes / time against the HBM roofline
        rows = [k for k in kern if k.startswith(HBM_CLASS["labels"]) and kern[k]["gbps_algorithmic"]]
        if rows and classes:
            ms = sum(kern[k]["ms_per_step"] for k in rows)
            fl = sum((kern[k]["tflops"] or 0.0) * kern[k]["ms_per_step"] for k in rows)
            nl = sum(kern[k]["launches_per_step"] for k in rows)
            by = sum(kern[k]["gbps_algorithmic"] * kern[k]["ms_per_step"] * 1e6 for k in rows)
            traffic, busy, src = pmc_class(HBM_CLASS["rocprof"], pmc_files) if pmc_files else (None, None, [])
            tbs = by / (ms * 1e-3) / 1e12
            classes["hbm_stream_conv"] = {
                "kernels": HBM_CLASS["what"], "instances": sorted(rows), "bound": "hbm", "gflop_per_step": round(fl, 1),
                "ms_per_step": round(ms, 4), "launches_per_step": nl, "avg_launch_us": round(1e3 * ms / nl, 2),
                "achieved": round(tbs, 3), "peak": HBM_PEAK_TBS, "unit": "TB/s", "frac": round(tbs / HBM_PEAK_TBS, 4),
                "achievable": HBM_ACHIEVABLE_TBS, "frac_of_achievable": round(tbs / HBM_ACHIEVABLE_TBS, 4),
                "tflops": round(fl / ms, 2), "pmc_mfma_busy": busy, "algorithmic_bytes_per_launch": round(by / nl),
                "traffic": traffic, "traffic_over_algorithmic": round(traffic / (by / nl), 2) if traffic and by else None,
                "pmc_source": src or None}
        roofline = None
        if classes:
            # `roofline` = the MATRIX class with the most time per step (the HBM stream class is reported next to them)
            dom = max((c for c in classes if classes[c].get("bound") != "hbm"), key=lambda c: classes[c]["ms_per_step"])
            d = classes[dom]
            tot_ms = sum(c["ms_per_step"] for c in classes.values())
            tot_fl = sum(c["gflop_per_step"] for c in classes.values())
            step_peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
            roofline = {"bound": "mfma", "kernel": "%s: %s" % (dom, d["kernels"]),
                        "achieved": d["achieved"], "peak": d["peak"], "unit": "TFLOP/s", "frac": d["frac"],
                        "traffic": d["traffic"], "traffic_source": d["pmc_source"],
                        "avg_launch_us": d["avg_launch_us"], "launches_per_step": d["launches_per_step"],
                        "measured": "hipEvents on the launch stream over %d serial steps (weight-gradient side stream "
                                    "off) right after the timed region; class = all launches of the listed kernel "
                                    "instances (same instances in every process: static tile table "
                                    "scouter_amd/tuning/gfx950.json)" % prof_steps,
                        "classes": classes,
                        "all_conv_kernels_tflops": round(tot_fl / tot_ms, 2),
                        # time-weighted fraction of each class's OWN matrix peak
                        "all_conv_kernels_frac": round(sum(c["frac"] * c["ms_per_step"] for c in classes.values()) / tot_ms, 4),
                        # the same FLOPs against the pipe the REFERENCE arithmetic would use (fp32 MFMA; round 1's
                        # yardstick): the bf16x3 plane kernels deliver fp32-grade products faster than the fp32 MFMA can
                        "all_conv_kernels_frac_of_step_peak": round(tot_fl / tot_ms / step_peak, 4),
                        "whole_step_frac": round(value * 3 * cfg["fwd_gflop"] * 1e9 / world / (step_peak * 1e12), 4)}
        metric = cfg["metric"].replace("224^2", "%d^2" % cfg["img_size"])
        line = {"metric": metric if cfg["batch"] == CONFIGS[a.config]["batch"] else
                metric.rsplit(", bs", 1)[0] + ", bs%d)" % cfg["batch"], "value": round(value, 2),
                "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
                "config": {"workload": "%s, %dx%d, per-GPU batch %d, random init, AdamW lr 1e-4%s"
                                       % (cfg["name"], cfg["img_size"], cfg["img_size"], cfg["batch"],
                                          "; backbone convolution matrix inputs in bf16, fp32 accumulate; the wide bottleneck "
                                          "tensors (block / conv3 / radix-conv outputs, residual-stream gradient) stored as "
                                          "bf16, statistics / sums / master weights / head fp32" if bf16 else ""),
                           "baseline_config": a.config, "precision": cfg["precision"],
                           "activation_storage": getattr(model, "activation_storage", "fp32"),
                           "register_split_bf16x3_layers": len([c for c in getattr(model, "_x3_convs", ()) if c.x3_mode()]),
                           "global_batch": world * cfg["batch"], "parallelism": "dp%d" % world,
                           "step": "zero_grad+fwd+loss+bwd(+allreduce)+adamw",
                           "launch": "hipGraph replay (one captured graph per step)" if graphed is not None else "eager",
                           "final_loss": round(loss_val, 5),
                           # diagnostic: wall time until the host had enqueued the K timed steps, per step -- well under
                           # ms_per_step: the GPU is the limiter; close to it: the eager launch chain is (slow / shared host)
                           "host_enqueue_ms_per_step": round(1e3 * t_host / a.steps, 3)},
                "roofline": roofline, "kernels": kern}
        if pmc_files and world == 1:
            hb = hbm_step_roofline(pmc_files, 1e3 * dt / a.steps)
            if hb is not None:
                line["hbm_roofline_step"] = hb
        if dp_info is not None:
            line["data_parallel"] = dp_info
        if world == 1 and not a.no_prof:
            line["xslot_roofline"] = xslot_roofline(device)
            # the reference's default geometry (--img_size 260 -> 9 x 9 = 81 tokens, reference train.py:39)
            line["xslot_roofline_n81"] = xslot_roofline(device, tokens=81)
            # the head of the configuration the metric is quoted on: 70 images x 10 slots
            line["xslot_metric_head"] = xslot_roofline(device, batch=70, slots=10, spc=1)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
