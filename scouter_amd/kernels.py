"""Tensor-level wrappers over the C ABI (include/scouter_hip.h): shape checks, output allocation, raw pointers and
the current HIP stream.  PyTorch provides device memory and streams only -- every arithmetic op below runs in a
hand-written HIP kernel of libscouter_hip.so, and there is no CPU / ATen fallback (CPU tensors raise).

Layouts: activations NHWC (torch tensors of shape [B, H, W, C], contiguous); convolution weights are
nn.Parameters of LOGICAL shape (Cout, Cin/g, kh, kw) whose PHYSICAL storage is HWIO ([kh][kw][Cin/g][Cout])."""
import os

import torch

from . import _native

F32 = torch.float32
BF16 = torch.bfloat16
# storage-type bits of the typed entry points (include/scouter_hip.h SCOUTER_IO_*)
IO_X_BF16, IO_Y_BF16, IO_R_BF16 = 1, 2, 4
DGRAD_IO_ADDEND, DGRAD_IO_DY, DGRAD_IO_DX = 4, 8, 16          # (SCOUTER_DGRAD_IO_*; bits 1 / 2: BnBwdFuse.x_io)
_ws = {}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream
    object through several Python layers: ~8 us, 170 times per training step.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=F32):
    """dtype: the storage type the callee accepts, or a tuple of them (typed-storage entry points)."""
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("scouter_amd: %s is on %s -- the xSlot path runs on a HIP device only (no CPU fallback)"
                           % (name, t.device))
    if t.dtype != dtype and not (isinstance(dtype, tuple) and t.dtype in dtype):
        raise RuntimeError("scouter_amd: %s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("scouter_amd: %s must be contiguous" % name)


def _p(t):
    return None if t is None else t.data_ptr()


def workspace(nbytes, device):
    """Per-(device, stream) scratch, grown on demand.  Kernels on one stream are ordered, so they can share it; the
    weight-gradient side stream gets its own."""
    key = (device.type, device.index, _stream() if device.type == "cuda" else 0)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


_arrival = {}
ARRIVAL_SLOTS = 1 << 16
# SCOUTER_SLAB_FUSE=1: the split-K slabs are summed INSIDE the weight-gradient kernels by the last workgroup of a tile to
# arrive (csrc/conv_common.h slab_tile_finish) instead of a slab-sum launch per layer.  Built, bit-
# reproducible and green under the poisoned-workspace tests -- and measured 17-25 % SLOWER on the step (interleaved A/B on one
# box, 2 x 60 steps each: 4 113 / 4 128 img/s with the separate launches; 3 432 / 3 461 with write-through stores + sc1 loads,
# 3 121 / 3 122 with plain stores + agent-scope release / acquire fences): a layer has 4-64 output tiles, so 4-64 workgroups
# walk `splits` x 16 KB of uncached partials each (~100 us tails) while the 10-us slab_reduce launch it replaces spreads the
# same bytes over every CU.  Off by default; kept as a documented option.
SLAB_FUSE = os.environ.get("SCOUTER_SLAB_FUSE", "0") == "1"


def arrival_counters(device):
    """Per-(device, stream) buffer of zeroed 32-bit arrival counters for the split-K weight-gradient kernels: with it the
    LAST workgroup of an output tile sums the tile's slabs inside the kernel (include/scouter_hip.h `arrival`) -- no
    slab-sum launch.  The kernels leave it zero; it is never part of the shared scratch (which tests poison and other
    kernels overwrite)."""
    if not SLAB_FUSE:
        return None
    key = (device.type, device.index, _stream())
    a = _arrival.get(key)
    if a is None:
        a = _arrival[key] = torch.zeros(ARRIVAL_SLOTS, dtype=torch.int32, device=device)
    return a


def hwio(weight):
    """Physical HWIO view [kh, kw, Cin/g, Cout] of a logical-OIHW conv weight; relayouts (once) if needed."""
    v = weight.permute(2, 3, 1, 0)
    if not v.is_contiguous():
        with torch.no_grad():
            weight.data = weight.data.permute(2, 3, 1, 0).contiguous().permute(3, 2, 0, 1)
        v = weight.permute(2, 3, 1, 0)
    return v


def oihw_view(flat, cout, cin_g, kh, kw):
    """Logical (Cout, Cin/g, kh, kw) view over a flat HWIO buffer."""
    return flat.view(kh, kw, cin_g, cout).permute(3, 2, 0, 1)


def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


# ---------------------------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------------------------
# precision = "fp32" (exact fp32 MFMA, the parity path) or "bf16" (matrix inputs rounded to bf16, fp32 accumulation;
# tensors stay fp32 in memory; BASELINE configs[4] names bf16).  It is an ARGUMENT of every convolution call, carried
# by the layer object (nn_hip.Conv2d.precision, set per model by SlotModel) -- there is no module-global switch, so
# models of different precision, or a backward on the autograd thread next to another model's forward, cannot race.
BF16_MIN_PIXELS = 1024      # layers with fewer GEMM rows (the split-attention FCs on the pooled vector) stay in fp32
# Block tiles / weight-gradient plans per layer shape.  DEFAULT: a STATIC table (scouter_amd/tuning/gfx950.json, made
# offline on an MI355X by tools_dev/tune_table.py with the timing autotuner below and committed), so every process, every
# data-parallel rank and every profiler run launches the SAME kernel instance for a layer shape -- gradients are
# bit-reproducible across processes and the rocprofv3 rows under profiles/ are the instances bench.py times.  Shapes the
# table does not hold take the entry of the nearest batch size with otherwise equal geometry, else the library's static
# heuristic (-1).  SCOUTER_AUTOTUNE=1 brings back timing on first use (per process; SCOUTER_TUNE_RECORD=<file> writes
# what it chose, which is how the table is made); SCOUTER_AUTOTUNE=0 ignores the table too (library heuristics only).
AUTOTUNE = os.environ.get("SCOUTER_AUTOTUNE", "table")
_tile_cache = {}
_tile_cands = {}            # key -> legal candidates, recorded for tools_dev/tune_in_step.py
_tune_table = None
_tune_index = None
TUNE_TABLE_PATH = os.environ.get("SCOUTER_TUNE_TABLE") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                       "tuning", "gfx950.json")


def _key_str(key):
    return "|".join(str(int(k)) if isinstance(k, bool) else str(k) for k in key)


def _batch_pos(key):
    """Index of the batch entry in a tuning key: ("fwd", bf16, B, ...), ("dgrad+bn", n, addend, bf16, B, ...),
    ("pdgrad+bn", n, addend, nplanes, B, ...), ("pfwd" | "pdgrad" | "pwgrad", nplanes, B, ...),
    ("wgrad" | "dgrad", bf16, B, ...)."""
    return 4 if key[0] in ("dgrad+bn", "pdgrad+bn", "xdgrad+bn") else 2


def _load_tune_table():
    global _tune_table, _tune_index
    import json
    _tune_table, _tune_index = {}, {}
    if os.path.exists(TUNE_TABLE_PATH):
        with open(TUNE_TABLE_PATH) as f:
            doc = json.load(f)
        # the table is only meaningful on the architecture it was timed on: elsewhere the library heuristics
        arch = ""
        if torch.cuda.is_available():
            arch = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "").split(":")[0]
        if not arch or arch == doc.get("arch", arch):
            _tune_table = {k: int(v) for k, v in doc["choices"].items()}
    for ks, v in _tune_table.items():
        parts = ks.split("|")
        bp = _batch_pos(parts)
        _tune_index.setdefault("|".join(parts[:bp] + ["*"] + parts[bp + 1:]), []).append((int(parts[bp]), v))
    for lst in _tune_index.values():
        lst.sort()


def _table_choice(key, launch):
    """The committed choice for this layer shape (exact key, else the same geometry at the nearest batch size -- ties
    to the smaller batch), if it is a legal candidate here; else -1."""
    import math
    if _tune_table is None:
        _load_tune_table()
    ks = _key_str(key)
    v = _tune_table.get(ks)
    if v is None:
        bp = _batch_pos(key)
        near = _tune_index.get("|".join([_key_str(key[:bp]), "*", _key_str(key[bp + 1:])]))
        if near:
            v = min(near, key=lambda e: (abs(math.log(e[0] / float(key[bp]))), e[0]))[1]
    if v is None or v < 0 or not launch(v, dry=True):
        return -1
    return v


def _record_choice(key, choice):
    path = os.environ.get("SCOUTER_TUNE_RECORD")
    if not path:
        return
    import atexit
    import json
    if not hasattr(_record_choice, "pending"):
        _record_choice.pending = {}

        def flush():
            old = {}
            if os.path.exists(path):
                with open(path) as f:
                    old = json.load(f).get("choices", {})
            old.update(_record_choice.pending)
            with open(path, "w") as f:
                json.dump({"arch": "gfx950", "choices": dict(sorted(old.items()))}, f, indent=0)
        atexit.register(flush)
    _record_choice.pending[_key_str(key)] = int(choice)


TUNE_REPS = int(os.environ.get("SCOUTER_TUNE_REPS", "3"))


def _pick_tile(key, launch, candidates=(0, 1, 2, 3)):
    """Block-tile / plan choice per (mode, layer shape).  Default: the committed static table (see AUTOTUNE above).
    SCOUTER_AUTOTUNE=1: time the candidates once (hipEvents on the current stream, first call only) and cache the winner.
    Forward tiles give bit-identical results; weight-gradient plans and plane tile 5 change the summation order."""
    t = _tile_cache.get(key)
    if t is not None:
        return t
    _tile_cands[key] = tuple(c for c in candidates if c < 0 or launch(c, dry=True))
    if AUTOTUNE == "0":
        _tile_cache[key] = -1
        return -1
    if AUTOTUNE != "1":
        t = _table_choice(key, launch)
        if t >= 0 and t not in candidates:
            t = -1
        _tile_cache[key] = t
        return t
    best, best_ms = -1, None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()          # nothing else on the device while the candidates are timed (side-stream callers!)
    for cand in candidates:
        if not launch(cand, dry=True):
            continue
        launch(cand)
        ms = None
        for _ in range(2 if TUNE_REPS > 3 else 1):          # (offline table runs: best of two batches)
            ev0.record()
            for _ in range(TUNE_REPS):
                launch(cand)
            ev1.record()
            ev1.synchronize()
            m = ev0.elapsed_time(ev1)
            ms = m if ms is None else min(ms, m)
        if best_ms is None or ms < best_ms:
            best, best_ms = cand, ms
    _tile_cache[key] = best
    _record_choice(key, best)
    return best


class BnBwdFuse:
    """Hand-over between a BatchNorm's backward and the input-gradient kernel that PRODUCES its dy (csrc/conv_common.h
    BnBwdFuse): `mask` = ReLU sign bits of the activation whose gradient is being produced (None: no ReLU), `entries` =
    [(x, saved)] of the one or two BatchNorms fed by that gradient.  A producer that fuses the reductions into its
    epilogue stores the masked gradient, fills `parts` (one [rows, C, 2] fp64 tensor per entry) and `rows`;
    `BatchNorm2d.bwd(..., fused=fuse.ext(i))` then skips its reduction pass.  A producer that cannot (unsupported
    shape / kernel) leaves `parts` None and the BatchNorm backward runs its own passes: `applied` tells which."""
    __slots__ = ("mask", "entries", "parts", "rows")

    def __init__(self, mask, entries):
        assert 1 <= len(entries) <= 2
        self.mask, self.entries = mask, entries
        self.parts, self.rows = None, 0

    @property
    def applied(self):
        return self.parts is not None

    def ext(self, i):
        return (self.parts[i], self.rows)

    def alloc(self, rows, x_shape):
        for x, saved in self.entries:
            assert tuple(x.shape) == tuple(x_shape), (tuple(x.shape), tuple(x_shape))
            _chk(x, "BatchNorm input", (F32, BF16)); _chk(saved, "BatchNorm saved block")
        C = x_shape[-1]
        self.rows = rows
        self.parts = [torch.empty((rows, C, 2), dtype=torch.float64, device=self.entries[0][0].device)
                      for _ in self.entries]

    def args(self):
        """(relu_mask, x1, saved1, part1, x2, saved2, part2) of the *_bnbwd entry points"""
        x1, s1 = self.entries[0]
        x2, s2 = self.entries[1] if len(self.entries) > 1 else (None, None)
        return (_p(self.mask), _p(x1), _p(s1), _p(self.parts[0]), _p(x2), _p(s2),
                _p(self.parts[1]) if len(self.entries) > 1 else None)

    def x_io(self):
        """bit i: the BatchNorm input of entry i is STORED as bf16 (only the bf16-input kernels' epilogue reads those)"""
        return sum(1 << i for i, (x, _) in enumerate(self.entries) if x.dtype == BF16)


_NO_FUSE = (None,) * 7
# Which producers fuse -- decided by A/B runs of the real step (interleaved `bench.py --steps 120` runs on one box, three
# each): bit 1 = 1x1 kernels -> one BatchNorm, bit 2 = 1x1 kernels -> two BatchNorms (block output +
# downsample branch), bit 3 = fp32 3x3 kernels (stem, resnet18), bit 0 = bf16x3 plane kernels -> the BatchNorm in front
# of the 3x3 layer.  First measurement (plain epilogue): bits 1+2 +1.3 %, bit 3 -0.3 %, bit 0 -0.6 %.  With the epilogue
# that requests its operands at workgroup start (fp32 kernels): 6 -> 4 012, 14 -> 4 043, 7 -> 4 018, 15 -> 4 048 img/s:
# everything fuses.
BN_BWD_FUSE = int(os.environ.get("SCOUTER_BN_FUSE", "15"))


def _fuse_wanted(post, kh, plane_kernel=False):
    if post is None:
        return False
    bit = 1 if plane_kernel else (8 if kh > 1 else (2 if len(post.entries) == 1 else 4))
    return bool(BN_BWD_FUSE & bit)


def _tile_legal(ng, t):
    return (t == 0 and ng % 128 == 0) or (t in (1, 2) and ng % 64 == 0) or t == 3


def _pw_persist_legal(M, K_, N_, kh, kw, stride, pad, groups, plain):
    """tile 4 = persistent pointwise kernel with the weight tile resident in LDS (csrc/conv_pw_persist.h): 1x1 / stride 1 /
    groups 1, GEMM-K of 64 / 128 / 256 channels, 64-multiples of output columns, plain epilogue (forward: optional
    BatchNorm statistics; input gradient: optional addend), byte offsets within 31 bits."""
    return bool(plain and kh == 1 and kw == 1 and stride == 1 and pad == 0 and groups == 1 and K_ in (64, 128, 256) and
                N_ % 64 == 0 and 4 * M * max(K_, N_) < (1 << 31))


PWB_FWD = os.environ.get("SCOUTER_PWB_FWD", "1") != "0"
PWB_DGRAD = os.environ.get("SCOUTER_PWB_DGRAD", "1") != "0"


def xpw_fwd_eligible(cin, cout, kh, kw, stride, pad, groups, has_bias):
    """Short-K pointwise layers whose FORWARD runs on the persistent bf16x3 kernel (csrc/conv_pw_persist_x3.h, tile 5 of the
    fp32 forward): 64 / 128 / 256 input channels, 64-multiples of output channels, no bias.  A STATIC rule (the forward
    is a function of the layer shapes alone, whatever the batch); SCOUTER_X3 bit 5."""
    return bool(kh == 1 and kw == 1 and stride == 1 and pad == 0 and groups == 1 and not has_bias and
                cin in (64, 128, 256) and cout % 64 == 0)


def conv2d_fwd(x, w_hwio, bias=None, addend=None, stride=1, pad=0, groups=1, relu=False, bn_stats=False,
               precision="fp32", out_dtype=F32, tile=None):
    """bn_stats=True: the epilogue also produces the per-tile fp64 channel sums BatchNorm needs; returns
    (y, (partial, rows)) and `bn_fwd(..., stats=(partial, rows))` then skips its own statistics pass.
    tile=5: the persistent bf16x3 kernel (xpw_fwd_eligible layers, plain epilogue; other calls fall back to the table).
    Activation storage (precision "bf16" only, layers the bf16 kernel runs): x may be a bfloat16 tensor -- the values the
    kernel rounds an fp32 input to, so the product is the same bits -- and out_dtype=torch.bfloat16 stores y rounded (the
    statistics still come from the fp32 accumulators)."""
    _chk(x, "x", (F32, BF16)); _chk(w_hwio, "weight"); _chk(bias, "bias"); _chk(addend, "addend")
    B, H, W, Cin = x.shape
    kh, kw, cg, Cout = w_hwio.shape
    assert cg * groups == Cin, (x.shape, w_hwio.shape, groups)
    L = _native.lib()
    st = _stream()

    Ho, Wo = conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad)
    bf16 = precision == "bf16" and B * Ho * Wo >= BF16_MIN_PIXELS
    if (x.dtype != F32 or out_dtype != F32) and not bf16:
        raise RuntimeError("scouter_amd: bf16-stored activations are an option of the bf16-input kernels (precision "
                           "'bf16', >= %d output pixels); got precision %r, %d pixels" % (BF16_MIN_PIXELS, precision, B * Ho * Wo))
    y = torch.empty((B, Ho, Wo, Cout), dtype=out_dtype, device=x.device)
    io = (IO_X_BF16 if x.dtype == BF16 else 0) | (IO_Y_BF16 if out_dtype == BF16 else 0)
    wt = None
    if bf16:                      # W^T as bf16 [taps][Cout][Cin/groups], rebuilt per call (weights change every step)
        wt = torch.empty((kh * kw, Cout, cg), dtype=torch.bfloat16, device=x.device)
        _native.check(L.scouter_conv2d_weight_bf16t(_p(w_hwio), _p(wt), kh, kw, Cin, Cout, groups, st), "weight_bf16t")

    plain = bias is None and addend is None and not relu
    picking = [AUTOTUNE != "1"]       # the table lookup asks whether the SHAPE admits tile 4; this call's epilogue is checked below

    def launch(tile, dry=False, part=None):
        if dry:
            if tile == 4:
                return not bf16 and _pw_persist_legal(B * H * W, Cin, Cout, kh, kw, stride, pad, groups,
                                                      plain or picking[0])
            return _tile_legal(Cout // groups, tile)
        if bf16:
            _native.check(L.scouter_conv2d_fwd_bf16_io(_p(x), _p(wt), _p(bias), _p(addend), _p(y), _p(part), B, H, W, Cin,
                                                       Cout, kh, kw, stride, pad, groups, int(relu), tile, io, st),
                          "conv2d_fwd_bf16")
        else:
            _native.check(L.scouter_conv2d_fwd_f32(_p(x), _p(w_hwio), _p(bias), _p(addend), _p(y), _p(part), B, H, W,
                                                   Cin, Cout, kh, kw, stride, pad, groups, int(relu), tile, st),
                          "conv2d_fwd")
        return True

    # (typed storage shares the fp32-storage key: the table is tuned on that; half the bytes move no tile boundary far)
    # (the cached choice is a function of the layer shape alone -- the key has no epilogue in it, and a first call with
    #  bias / addend / ReLU must not cache "-1" for the plain calls of the same shape, ADVICE r4 -- legality of tile 4 for
    #  THIS call's epilogue is decided here, per call)
    if tile == 5 and not (plain and not bf16 and xpw_fwd_eligible(Cin, Cout, kh, kw, stride, pad, groups, False) and
                          (B * H * W + 128) * Cout < (1 << 29) and B * H * W * Cin < (1 << 29)):
        tile = None
    # bf16 mode, input STORED as bf16: the pointwise layers with up to 512 input channels run on the persistent typed kernel
    # (csrc/conv_pw_persist_bf16.h pwb_fwd_kernel; tile 4 of the typed forward) -- a static rule of shape and storage, never of
    # timing.  SCOUTER_PWB_FWD=0: the tile kernels.
    if (tile is None and bf16 and PWB_FWD and plain and x.dtype == BF16 and kh == 1 and kw == 1 and stride == 1 and pad == 0 and
            groups == 1 and Cin in (64, 128, 256, 512) and Cout % 128 == 0 and B * H * W * Cin < (1 << 30) and
            (B * H * W + 128) * Cout * (2 if out_dtype == BF16 else 4) < (1 << 32)):
        tile = 4
    if tile is None:
        tile = _pick_tile(("fwd", bf16, B, H, W, Cin, Cout, kh, kw, stride, pad, groups), launch, (0, 1, 2, 3, 4))
        picking[0] = False
        if tile == 4 and not launch(4, dry=True):    # (the table's entry is for the plain epilogue; this call has bias / addend / ReLU)
            tile = -1
        # bf16 mode, 3x3 / stride 1 / pad 1 with 32 input channels per group (deep stem, layer1's radix convolutions): the 128-pixel
        # tiles (1: 64 columns per group, 3: 32), behind which the library runs the persistent resident-rows kernel when x is STORED
        # as bf16 (csrc/conv_halo_dgrad_bf16.h bhalo_fwd_kernel) -- bit-identical to the tile kernels, output and statistics rows,
        # so the rule names the tile by the SHAPE (behind the table's lookup, whose entry it overrides) and storage stays bit-neutral.
        if (bf16 and BHALO and kh == 3 and kw == 3 and stride == 1 and pad == 1 and cg == 32 and
                Cout // groups in (32, 64) and W <= 112 and H >= 2):
            tile = 1 if Cout // groups == 64 else 3
    part, rows = None, 0
    if bn_stats:
        rows_fn = L.scouter_conv2d_fwd_bn_partial_rows_bf16 if bf16 else L.scouter_conv2d_fwd_bn_partial_rows
        rows = rows_fn(B, H, W, Cin, Cout, kh, kw, stride, pad, groups, tile)
        part = torch.empty((rows, Cout, 2), dtype=torch.float64, device=x.device)
    launch(tile, part=part)
    return (y, (part, rows)) if bn_stats else y


def conv2d_dgrad(dy, w_hwio, x_shape, addend=None, stride=1, pad=0, groups=1, precision="fp32", post=None,
                 out_dtype=F32):
    """post (BnBwdFuse): the tensor produced is the gradient of a BatchNorm+ReLU block output -- the epilogue masks it
    and reduces that BatchNorm's backward sums (see BnBwdFuse).
    Gradient storage (bf16-input kernel only): dy may be a bfloat16-stored tensor (the values the kernel rounds an fp32
    dy to: same result), the addend too; out_dtype=torch.bfloat16 stores dx rounded WHEN the fused epilogue runs (dx is
    then the masked block-output gradient, read by that BatchNorm's backward and as the previous block's addend) -- check
    dx.dtype."""
    _chk(dy, "dy", (F32, BF16)); _chk(w_hwio, "weight"); _chk(addend, "addend", (F32, BF16))
    B, H, W, Cin = x_shape
    kh, kw, cg, Cout = w_hwio.shape
    L = _native.lib()
    st = _stream()

    # strided input gradients (resnet18) and tiny layers stay on the fp32 kernel
    bf16 = precision == "bf16" and stride == 1 and B * H * W >= BF16_MIN_PIXELS
    # (BatchNorm inputs stored as bf16: only the bf16-input kernel's epilogue reads them; otherwise that BatchNorm's own
    #  backward -- typed -- does the reduction)
    fuse = _fuse_wanted(post, kh) and (bf16 or post.x_io() == 0)
    if (BHALO and bf16 and dy.dtype == F32 and kh == 3 and kw == 3 and pad == 1 and Cin // groups == 32 and
            Cout // groups in (32, 64) and W <= 112 and H >= 2 and B * H * W * Cout * 2 < (1 << 31) - (1 << 20)):
        # 3x3 layers with 32 input channels per group: the library's resident-rows input gradient (csrc/conv_halo_dgrad_bf16.h)
        # reads a bf16-STORED dy.  The rule is a function of the shape only -- an fp32-stored dy (SCOUTER_BF16_GRADS=0) is rounded
        # to bf16 (RNE: the value the tile kernel's loader rounds it to) by one extra pass first: storage stays bit-neutral
        dy = planes_split(dy, 1)[0]
    if out_dtype == BF16 and not (fuse and bf16):
        out_dtype = F32           # (bf16 storage is for the MASKED block-output gradient the fused epilogue writes)
    dx = torch.empty(x_shape, dtype=out_dtype, device=dy.device)
    io = ((DGRAD_IO_DY if dy.dtype == BF16 else 0) | (DGRAD_IO_DX if out_dtype == BF16 else 0) |
          (DGRAD_IO_ADDEND if addend is not None and addend.dtype == BF16 else 0))
    if io and not bf16:
        raise RuntimeError("scouter_amd: bf16-stored gradients are an option of the bf16-input kernels (precision 'bf16', "
                           "stride 1, >= %d pixels); got precision %r, stride %d, %d pixels"
                           % (BF16_MIN_PIXELS, precision, stride, B * H * W))

    def launch(tile, dry=False, fuse=_NO_FUSE):
        if dry:
            if tile == 4 and bf16:
                # the persistent typed kernel (csrc/conv_pw_persist_bf16.h): fused launches with every tensor stored as bf16
                return bool(will_fuse and kh == 1 and kw == 1 and pad == 0 and groups == 1 and Cout in (64, 128, 256, 512) and
                            Cin % 128 == 0 and (B * H * W + 128) * Cin < (1 << 30) and B * H * W * Cout < (1 << 30) and
                            (io & DGRAD_IO_DY) and (io & DGRAD_IO_DX) and (addend is None or io & DGRAD_IO_ADDEND) and
                            post.x_io() == (1 << len(post.entries)) - 1)
            if tile == 4:        # (GEMM-K of the input gradient = Cout)
                return _pw_persist_legal(B * H * W, Cout, Cin, kh, kw, stride, pad, groups, True)
            if tile == 5:
                # the persistent bf16x3 kernel (csrc/conv_pw_persist_x3.h): fused fp32 launches of the short-K 1x1 layers
                return bool(will_fuse and not bf16 and kh == 1 and kw == 1 and stride == 1 and pad == 0 and groups == 1 and
                            Cout in (64, 128, 256) and Cin % 64 == 0 and (B * H * W + 128) * Cin < (1 << 29) and
                            B * H * W * Cout < (1 << 29))
            return _tile_legal(Cin // groups, tile)
        if bf16:
            _native.check(L.scouter_conv2d_dgrad_bnbwd_bf16_io(
                _p(dy), _p(w_hwio), _p(addend), _p(dx), B, H, W, Cin, Cout, kh, kw, stride, pad, groups, tile, *fuse,
                io | (post.x_io() if fuse is not _NO_FUSE else 0), st), "conv2d_dgrad")
        else:
            _native.check(L.scouter_conv2d_dgrad_bnbwd_f32(
                _p(dy), _p(w_hwio), _p(addend), _p(dx), B, H, W, Cin, Cout, kh, kw, stride, pad, groups, tile, *fuse, st),
                "conv2d_dgrad")
        return True

    will_fuse = fuse
    if fuse:
        # the fused launch is tuned as what it is (its epilogue reads one or two more tensors: on the short-K layers a
        # different tile wins than for the plain input gradient); partials sized for the most rows while timing
        def launch_fused(tile, dry=False):
            if dry:
                return launch(tile, dry=True)
            if not post.applied:
                post.alloc(max(-(-B * H * W // 64), 512), x_shape)   # (the most rows any tile writes; tile 4: <= 512)
            return launch(tile, fuse=post.args())
        tile = _pick_tile(("dgrad+bn", len(post.entries), addend is not None, bf16, B, H, W, Cin, Cout, kh, kw, stride,
                           pad, groups), launch_fused, (0, 1, 2, 3, 4, 5))
        if tile < 0:                              # autotuning disabled: name a tile, the partial rows depend on it
            tile = 2 if (Cin // groups) % 64 == 0 else 3
        if bf16:
            rows = L.scouter_conv2d_dgrad_bn_partial_rows_bf16(B, H, W, Cin, Cout, kh, kw, stride, pad, groups, tile,
                                                               int(len(post.entries) > 1))
        else:
            rows = L.scouter_conv2d_dgrad_bn_partial_rows(B, H, W, Cin, Cout, kh, kw, stride, pad, groups, tile)
        post.alloc(rows, x_shape)
        launch(tile, fuse=post.args())
    elif (bf16 and PWB_DGRAD and kh == 1 and kw == 1 and pad == 0 and groups == 1 and Cout in (64, 128, 256, 512) and
          Cin % 64 == 0 and (io & DGRAD_IO_DY) and not (io & DGRAD_IO_DX) and (B * H * W + 128) * Cin < (1 << 30) and
          B * H * W * Cout < (1 << 30)):
        # bf16 mode, dy STORED as bf16: the plain pointwise input gradients (conv3 / downsample) run on the persistent typed
        # kernel (csrc/conv_pw_persist_bf16.h pwb_dgrad_kernel) -- a static rule of shape and storage (Cout <= 512: at 1024 the
        # tile kernel is level, tools_dev/pwb_dgrad_bench.py: 64 <- 256 at 56 x 56 141 -> 115 us, 64 <- 64 87 -> 49, 128 <- 512
        # at 28 x 28 91 -> 76; +0.15 % on config 5, most of whose plain input gradients read an fp32 dy).  SCOUTER_PWB_DGRAD=0:
        # the tile kernels
        launch(4)
    else:
        launch(_pick_tile(("dgrad", bf16, B, H, W, Cin, Cout, kh, kw, stride, pad, groups), launch, (0, 1, 2, 3, 4)))
    return dx


# ---- convolution on pre-split bf16 operand planes (csrc/conv_planes.hip)


class PlaneTensor:
    """An activation as `planes` ([np, B, H, W, C] bf16, the operand of the plane convolutions) and, unless every
    consumer runs on planes, also as `f32` (NHWC fp32; None otherwise).  Produced by bn_fwd(planes=np) and
    sa_bn_bwd(planes=np)."""
    __slots__ = ("f32", "planes")

    def __init__(self, f32, planes):
        self.f32, self.planes = f32, planes

    @property
    def shape(self):
        return self.planes.shape[1:]

    @property
    def device(self):
        return self.planes.device


def planes_split(x, nplanes=3):
    """fp32 tensor -> [nplanes, *x.shape] bf16 planes with x == sum(planes) exactly for nplanes = 3."""
    _chk(x, "x")
    out = torch.empty((nplanes,) + tuple(x.shape), dtype=BF16, device=x.device)
    _native.check(_native.lib().scouter_planes_split_f32(_p(x), _p(out), x.numel(), nplanes, _stream()), "planes_split")
    return out


def planes_split_weight(w_hwio, groups, nplanes=3, fwd=True, dgrad=True):
    """HWIO fp32 weight -> (forward planes [np, taps, Cout, Cin/g], dgrad planes [np, taps, Cin, Cout/g])."""
    kh, kw, cg, Cout = w_hwio.shape
    Cin = cg * groups
    wf = torch.empty((nplanes, kh * kw, Cout, cg), dtype=BF16, device=w_hwio.device) if fwd else None
    wd = torch.empty((nplanes, kh * kw, Cin, Cout // groups), dtype=BF16, device=w_hwio.device) if dgrad else None
    _native.check(_native.lib().scouter_planes_split_weight_f32(_p(w_hwio), _p(wf), _p(wd), kh, kw, Cin, Cout, groups,
                                                                nplanes, _stream()), "planes_split_weight")
    return wf, wd


class PlaneWeightSplitter:
    """Splits the weights of every plane convolution of a model in ONE launch per step into persistent plane buffers (the
    weights change in place with every optimizer step; pointers, table and buffers are set up once) -- one 16-us
    `split_weight_kernel` launch per layer was 0.13 ms per step."""

    def __init__(self):
        # one entry per (planes, device, weights, wanted outputs) key, kept for the life of the model: a training forward
        # (forward + input-gradient planes) and an eval forward (forward planes only) have different keys, and a captured
        # hipGraph has the table / plane-buffer addresses of ITS key baked in -- replacing the single cached entry freed
        # buffers under the graph's replays.  Alternating train / eval also no longer rebuilds the table.
        self._entries = {}

    def run(self, items, nplanes):
        """items: [(w_hwio, groups, want_fwd, want_dgrad)] -> [(wf, wd)]"""
        import struct
        L = _native.lib()
        dev = items[0][0].device
        key = (nplanes, dev, tuple((w.data_ptr(), g, f, d) for w, g, f, d in items))
        ent = self._entries.get(key)
        if ent is None:
            rows, outs, first = [], [], 0
            for w, groups, fwd, dgrad in items:
                _chk(w, "weight")
                kh, kw, cg, Cout = w.shape
                if cg % 32 or Cout % 32 or (Cout // groups) % 32:
                    # (the tiled kernel walks [32 ci][32 co] tiles and assumes every tensor starts on a 1024-element boundary)
                    raise RuntimeError("scouter_amd: the one-launch weight split takes 32-multiples of channels per group, got "
                                       "a %dx%d weight with %d input channels per group, %d output channels, %d groups"
                                       % (kh, kw, cg, Cout, groups))
                wf = torch.empty((nplanes, kh * kw, Cout, cg), dtype=BF16, device=dev) if fwd else None
                wd = torch.empty((nplanes, kh * kw, cg * groups, Cout // groups), dtype=BF16, device=dev) if dgrad else None
                rows.append(struct.pack("<QQQiiiiq", w.data_ptr(), _p(wf) or 0, _p(wd) or 0, kh * kw, cg, Cout, groups, first))
                outs.append((wf, wd))
                first += kh * kw * cg * Cout
            blob = b"".join(rows)
            assert len(blob) == len(rows) * L.scouter_planes_split_weights_row_bytes()
            ent = self._entries[key] = (torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev), outs, first)
        table, outs, total = ent
        _native.check(L.scouter_planes_split_weights_multi(_p(table), len(outs), total, nplanes, _stream()),
                      "planes_split_weights_multi")
        return outs

    def buffers(self):
        """Every device buffer of every entry (GraphedTrainStep keeps them alive next to the workspaces)."""
        return [t for table, outs, _ in self._entries.values() for t in [table] + [b for o in outs for b in o if b is not None]]


_PLANE_TILE_ROWS = {0: 128, 1: 128, 2: 128, 3: 64, 4: 256, 5: 256, 6: 64}


def _plane_part_rows(tile, M):
    """Partial rows the plane kernels write for fused BatchNorm sums: one per M tile -- tile 6 (persistent 256-row tiles)
    one per 64-row WAVE row of every tile, 4 per tile, incl. the all-zero ones beyond M."""
    if tile == 6:
        return (M + 255) // 256 * 4
    return (M + _PLANE_TILE_ROWS[tile] - 1) // _PLANE_TILE_ROWS[tile]


def _plane_tiles(ng, nplanes=3, halo=False):
    """Block tiles of the plane kernels (csrc/conv_planes.hip dispatch_pconv): 0 = 128x128, 1 = 128x64 (three LDS
    stages), and for three planes 2 = 128x64 (two workgroups per CU), 3 = 64x64 (three), 4 = 256x128 with eight waves,
    5 (halo=True: same-size 3x3 layers on maps up to 63 wide) = 256 x (128 | 64) with the input rows resident in LDS for
    all nine taps, 6 = PERSISTENT 256 x (128 | 64): one workgroup per CU walks the tile list, the DMA stream prefetches
    the next tile under the current one, register epilogue (csrc/conv_planes_persist.h).  Tiles 0-4 and 6 sum each output
    ELEMENT in the same order (bit-identical convolution outputs; the fused BatchNorm partial sums are grouped per tile,
    so statistics agree between tiles to fp64 rounding, not bit for bit); tile 5 sums over K in a different order
    (16-channel chunk outer, tap inner): equal to fp32 rounding."""
    if nplanes == 1:
        return ((0, 1) if ng % 128 == 0 else (1,)) + ((5,) if halo else ()) + (6,)
    return ((0, 2, 3, 4) if ng % 128 == 0 else (2, 3)) + ((5,) if halo else ()) + (6,)


# Plane tile 5 (input rows resident in LDS) sums K in another order than tiles 0-4.  bit 1: the input gradient may use it
# (autotuned against the others: gradients depend on autotuned summation orders anyway).  bit 0: the FORWARD uses it by a
# STATIC rule -- every same-size 3x3 layer on a map of at least 14 x 14 pixels, whatever the batch -- never by timing: a
# forward whose bits depended on the autotuner's choice differed between batch sizes / processes / data-parallel ranks by
# 1e-7, enough to flip a ReLU here and there (tests/test_model_gpu.py, the batch-70 half-batch property).  With the
# static rule the forward is a function of the layer shapes alone.  +0.3 % (bit 1) / +0.5 % (bit 0) images/sec.
# Default since round 6: 3 (both).  Round 5 kept bit 0 off because on ONE draw (the full-size parity fixture, seed 200 against the
# reference's 8-thread fp32 run) the forward with tile 5 lands 1.84 x PyTorch-fp32's own deviation from fp64 (tiles 0-4: 1.27 x;
# that test's bound then: 1.5 x).  The yardstick is a distribution now -- five seeds x the reference at three thread counts,
# tests/test_model_gpu.py::test_rounding_noise_over_five_seeds / ::test_gradient_noise_over_five_seeds: geometric mean of the ratio
# <= 1.5, no seed > 3, north_star's max(1e-4, 3 x floor) gate per seed -- and tile 5 sits inside it like tiles 0-4 (measured
# geometric means 0.53-0.75 forward, 0.27-0.77 head gradients for every option set).  SCOUTER_HALO=2: round 5's default.
HALO_TILE = int(os.environ.get("SCOUTER_HALO", "3"))
HALO_FWD_MIN_PIXELS = 196


def _halo_ok(kh, kw, stride, pad, H, W, mode=3):
    return bool(HALO_TILE & mode) and kh == 3 and kw == 3 and stride == 1 and pad == 1 and W <= 63


def conv2d_fwd_planes(xp, wf, kh, kw, stride=1, pad=0, groups=1, bias=None, addend=None, relu=False, bn_stats=False,
                      tile=None, out_dtype=F32):
    """xp: activation planes [np, B, H, W, Cin]; wf: forward weight planes.  Returns y (NHWC) or (y, stats).
    tile None: autotuned once per layer shape (bit-identical results for every tile).  out_dtype=torch.bfloat16: y is
    stored as bf16 (the statistics still come from the fp32 accumulators)."""
    nplanes, B, H, W, Cin = xp.shape
    Cout = wf.shape[2]
    y = torch.empty((B, conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad), Cout), dtype=out_dtype, device=xp.device)
    L = _native.lib()
    cands = _plane_tiles(Cout // groups, nplanes, False)
    if out_dtype == BF16:
        cands = tuple(t for t in cands if t != 6)      # (the persistent tile's register epilogue writes fp32 only)
    if tile is None and _halo_ok(kh, kw, stride, pad, H, W, 1) and H * W >= HALO_FWD_MIN_PIXELS:
        tile = 5                                   # static rule (see HALO_TILE): no timing, same bits for every batch
    if tile == 5:
        cands = cands + (5,)
    M = y.numel() // Cout
    if (bn_stats and (bias is not None or addend is not None)) or kh * kw * (Cin // groups // 32) < 2:
        cands = tuple(t for t in cands if t != 6)      # (tile 6 takes the statistics from the raw accumulators and needs
        #                                                  two K-tiles of 32 channels x taps for its DMA pipeline)
    # scratch for the statistics of the largest partial count (64-row tiles) while the tiles are being timed
    scratch = [None]

    def launch(t, dry=False, part=None):
        if dry:
            return t in cands
        if bn_stats and part is None:
            if scratch[0] is None:
                scratch[0] = torch.empty(((M + 255) // 256 * 4, Cout, 2), dtype=torch.float64, device=xp.device)
            part = scratch[0]
        _native.check(L.scouter_conv2d_fwd_planes_io(_p(xp), _p(wf), _p(bias), _p(addend), _p(y), _p(part), B, H, W, Cin,
                                                     Cout, kh, kw, stride, pad, groups, int(relu), nplanes, t,
                                                     IO_Y_BF16 if out_dtype == BF16 else 0, _stream()),
                      "conv2d_fwd_planes")
        return True
    if tile is None:
        tile = _pick_tile(("pfwd", nplanes, B, H, W, Cin, Cout, kh, kw, stride, pad, groups), launch, cands)
        if tile < 0:
            tile = cands[0]
    part, rows = None, 0
    if bn_stats:
        rows = _plane_part_rows(tile, M)                                       # one partial per M tile of the kernel
        part = torch.empty((rows, Cout, 2), dtype=torch.float64, device=xp.device)
    launch(tile, part=part)
    return (y, (part, rows)) if bn_stats else y


def conv2d_dgrad_planes(dyp, wd, x_shape, kh, kw, stride=1, pad=0, groups=1, addend=None, tile=None, post=None):
    nplanes = dyp.shape[0]
    B, H, W, Cin = x_shape
    Cout = dyp.shape[-1]
    dx = torch.empty(x_shape, dtype=F32, device=dyp.device)
    cands = _plane_tiles(Cin // groups, nplanes, _halo_ok(kh, kw, stride, pad, H, W, 2))
    if kh * kw * (Cout // groups // 32) < 2:
        cands = tuple(t for t in cands if t != 6)      # (tile 6 needs two K-tiles of 32 channels x taps)

    def launch(t, dry=False, fuse=_NO_FUSE):
        if dry:
            return t in cands
        _native.check(_native.lib().scouter_conv2d_dgrad_planes_bnbwd(
            _p(dyp), _p(wd), _p(addend), _p(dx), B, H, W, Cin, Cout, kh, kw, stride, pad, groups, nplanes, t, *fuse,
            _stream()), "conv2d_dgrad_planes")
        return True
    fused = _fuse_wanted(post, kh, True) and nplanes == 3  # (one-plane kernels, bf16 mode: measured -0.6 % on config 5)
    if tile is None and fused:
        # the fused launch is tuned as what it is (its epilogue masks the gradient, reads the BatchNorm input(s) and
        # reduces their sums: the persistent tile's register epilogue and the LDS-staged one cost differently)
        def launch_fused(t, dry=False):
            if dry:
                return t in cands
            if not post.applied:
                post.alloc((B * H * W + 255) // 256 * 4, x_shape)         # (the most rows any tile writes)
            return launch(t, fuse=post.args())
        tile = _pick_tile(("pdgrad+bn", len(post.entries), addend is not None, nplanes, B, H, W, Cin, Cout, kh, kw, stride,
                           pad, groups), launch_fused, cands)
        if tile < 0:
            tile = None
    if tile is None:
        tile = _pick_tile(("pdgrad", nplanes, B, H, W, Cin, Cout, kh, kw, stride, pad, groups), launch, cands)
        if tile < 0:
            tile = cands[0]
    if fused:
        post.alloc(_plane_part_rows(tile, B * H * W), x_shape)
        launch(tile, fuse=post.args())
    else:
        launch(tile)
    return dx


# ---- convolutions on the bf16 matrix cores, operand split in registers (csrc/conv_x3.hip)
# Which layers take it is a STATIC function of the layer's shape (never of timing or of the batch): the forward stays a
# function of the layer shapes, whatever the batch size / process / data-parallel rank.  Measured per layer
# (tools_dev/x3_bench.py): the 1x1 layers with Cin * Cout >= 2^16 -- every 1x1 of layer2-4 -- gain, the short-K 56 x 56
# layers are HBM-co-bound and stay on the persistent fp32 kernel; of the 3x3 layers those with 32 channels per group (the
# deep stem, the first radix convolution) -- the plane kernels need 64 -- ran on the fp32 MFMA kernels until round 5.
# SCOUTER_X3 bits: 0 forward, 1 plain input gradient, 2 input gradient with the fused BatchNorm-backward epilogue -- only
# where the GEMM is deep enough to carry it (K = Cout >= X3_FUSED_MIN_K: 149 -> 121 us, 78 -> 63 us; the K = 128 / 256
# launches are bound by the epilogue's streams and stay on the fp32 kernel that prefetches them), 3 weight gradient (1x1),
# 4 the forward of the 3x3 layers of x3_conv_eligible; 0: everything on the fp32 MFMA kernels.
# Bit 4: +1 % images/sec (4 455 vs 4 412).  Bit 5 (round 5, late): the forward of the short-K pointwise layers (64 / 128 input
# channels; 256 where the register-split GEMM does not serve the layer) on the PERSISTENT bf16x3 kernel
# (csrc/conv_pw_persist_x3.h, tile 5 of the fp32 forward) -- output streams the fp32 kernels run at 2.7 TB/s: 64 -> 256 at
# 56 x 56 109 -> 76 us, 128 -> 512 at 28 x 28 82 -> 53 us (tools_dev/xpw_fwd_bench.py), +1.1 % images/sec; a third of the fp32
# kernel's rounding error per layer.
# Default since round 6: 63 (everything).  Round 5 kept bits 4 / 5 off because ONE draw of the head's amplified rounding noise
# (seed 200 of the full-size parity fixture against the reference's 8-thread fp32 run) put them over that test's single-draw
# bounds -- head gradients at 2.5 x PyTorch-fp32's own deviation with bit 4 (bound 2 x), log-probabilities at 2.1 x with bit 5
# (bound 1.5 x) -- although the kernels are bit-identical to the plane kernels and a third of the fp32 kernel's error.  The
# yardstick is a DISTRIBUTION now (five seeds x the reference at three thread counts, forward and gradients:
# tests/test_model_gpu.py::test_rounding_noise_over_five_seeds, ::test_gradient_noise_over_five_seeds; generator
# oracle/gen_golden.py): every option set, this default included, has a geometric-mean ratio of 0.5-0.8 (bound 1.5) and no seed
# over 3 (tools_dev/seed_noise.py prints the table); north_star's max(1e-4, 3 x floor) gate holds per seed.  SCOUTER_X3=15:
# round 5's default.
# Bits 6 / 7 (round 6): the 3x3 passes whose GEMM is 32 columns wide per group on the persistent resident-rows kernel with the
# split in registers (csrc/conv_xhalo.hip, x3_halo_eligible) -- bit 6 the INPUT GRADIENTS (both deep-stem convolutions, layer1's
# radix convolutions: 220 -> 138, 366 -> 233, 2 x 187 -> 112 us with the fused BatchNorm-backward epilogue at batch 70; +3 %
# images/sec), bit 7 the FORWARD of the stem's 32 -> 32 convolution (191 -> 113 us, +0.5 %).  Default 127: bit 7 is OFF.  The
# kernel is a third of the fp32 kernel's rounding error per element, yet with it in the forward the five-seed distribution reads
# [3.98, 0.61, 0.90, 0.71, 1.55] (geometric mean 1.19 <= 1.5, but seed 200 is over the per-seed cap of 3: 2.6e-4 from fp64
# against the reference's own 4.0e-5 ... 1.06e-4) -- the criterion decides, not the kernel's merit: the head at random
# initialisation amplifies ANY change of summation order chaotically.  Input gradients do not enter the forward at all.
X3_DEFAULT = int(os.environ.get("SCOUTER_X3", "127"))
X3_FUSED_MIN_K = 512
X3_MIN_CHANNEL_PRODUCT = 1 << 16
_X3_TILES = (0, 1, 2, 3, 4, 5, 6)


def x3_eligible(cin, cout, kh, kw, stride, pad, groups, has_bias):
    """Pointwise layers the register-split bf16x3 GEMM serves, forward AND input gradient (both GEMM widths 64-multiples)."""
    return bool(kh == 1 and kw == 1 and stride == 1 and pad == 0 and groups == 1 and not has_bias and
                cin % 64 == 0 and cout % 64 == 0 and cin * cout >= X3_MIN_CHANNEL_PRODUCT)


def x3_conv_eligible(cin, cout, kh, kw, stride, pad, groups, has_bias):
    """3x3 / stride 1 / pad 1 layers with 32 input channels per group and 64-multiples of output channels per group: the
    FORWARD runs on the register-split kernel (the stem's 32 -> 64 convolution: 308 -> 233 us at batch 70).  Measured and
    left on the fp32 MFMA kernels (tools_dev/tune_x3.py): everything with 32 OUTPUT columns per group -- the 32 -> 32 stem
    convolution and every input gradient of these layers -- where one split (6.5 VALU per element) feeds a single 32-column
    MFMA block: 8 VALU per MFMA, 0.64-0.77x the fp32 kernel."""
    return bool(kh == 3 and kw == 3 and stride == 1 and pad == 1 and not has_bias and cin % groups == 0 and
                cout % groups == 0 and cin // groups == 32 and (cout // groups) % 64 == 0)


def x3_halo_eligible(cin, cout, kh, kw, stride, pad, groups, has_bias):
    """(forward, input gradient): which passes of a 3x3 / stride 1 / pad 1 layer run on the PERSISTENT resident-rows kernel with
    the split in registers (csrc/conv_xhalo.hip; tile 7 of the register-split entry points; SCOUTER_X3 bits 7 / 6) -- the passes whose
    GEMM is 32 columns wide per group: the forward with 32 output channels per group (the stem's 32 -> 32), the input gradient
    with 32 input channels per group (both deep-stem convolutions, layer1's radix convolution).  A static function of the
    layer's channels; the map width (<= 126) is checked per call."""
    if not (kh == 3 and kw == 3 and stride == 1 and pad == 1 and cin % groups == 0 and cout % groups == 0):
        return False, False
    cg, ng = cin // groups, cout // groups
    return bool(ng == 32 and cg % 32 == 0 and not has_bias), bool(cg == 32 and ng % 32 == 0)


X3_HALO_MAX_W = 126


def _x3_tile_ok(t, n):
    if t == 7:
        return n == 32
    return n % 128 == 0 if t in (0, 1) else (n % 64 == 0 if t in (2, 3, 4) else (n % 32 == 0 and t in (5, 6)))


def conv2d_fwd_x3(x, wf, addend=None, relu=False, bn_stats=False, tile=None, kh=1, pad=0, groups=1):
    """x: fp32 NHWC [B, H, W, Cin]; wf: forward weight planes [3, kh*kh, Cout, Cin/groups] (planes_split_weight).  Returns
    y or (y, (partial, rows)) like conv2d_fwd.  Every tile gives the same bits; the choice comes from the static table."""
    _chk(x, "x"); _chk(addend, "addend")
    B, H, W, Cin = x.shape
    Cout = wf.shape[2]
    assert wf.dtype == BF16 and wf.shape[0] == 3 and wf.shape[1] == kh * kh and wf.shape[3] * groups == Cin, \
        (tuple(wf.shape), tuple(x.shape))
    L = _native.lib()
    y = torch.empty((B, H, W, Cout), dtype=F32, device=x.device)
    M, Ng = B * H * W, Cout // groups
    scratch = [None]

    def launch(t, dry=False, part=None):
        if dry:
            return _x3_tile_ok(t, Ng)
        if bn_stats and part is None:
            if scratch[0] is None:
                scratch[0] = torch.empty((max((M + 63) // 64, 512), Cout, 2), dtype=torch.float64, device=x.device)
            part = scratch[0]
        _native.check(L.scouter_conv2d_fwd_x3(_p(x), _p(wf), None, _p(addend), _p(y), _p(part), B, H, W, Cin, Cout, kh, kh,
                                              pad, groups, int(relu), t, _stream()), "conv2d_fwd_x3")
        return True
    if tile is None:
        key = ("xfwd", 3, B, H, W, Cin, Cout) if kh == 1 and groups == 1 else ("xfwd", 3, B, H, W, Cin, Cout, kh, groups)
        tile = _pick_tile(key, launch, _X3_TILES)
    tile = L.scouter_conv2d_x3_tile(M, Ng, tile)
    part, rows = None, 0
    if bn_stats:
        rows = L.scouter_conv2d_x3_halo_partial_rows(groups) if tile == 7 else L.scouter_conv2d_x3_partial_rows(M, Ng, tile)
        part = torch.empty((rows, Cout, 2), dtype=torch.float64, device=x.device)
    launch(tile, part=part)
    return (y, (part, rows)) if bn_stats else y


def conv2d_dgrad_x3(dy, wd, x_shape, addend=None, post=None, tile=None, kh=1, pad=0, groups=1):
    """dy: fp32 NHWC [B, H, W, Cout]; wd: input-gradient weight planes [3, kh*kh, Cin, Cout/groups].  post (BnBwdFuse): as
    conv2d_dgrad -- fused when SCOUTER_BN_FUSE wants it and every BatchNorm input is fp32."""
    _chk(dy, "dy"); _chk(addend, "addend")
    B, H, W, Cin = x_shape
    Cout = dy.shape[-1]
    assert wd.dtype == BF16 and wd.shape[0] == 3 and wd.shape[1] == kh * kh and wd.shape[2] == Cin and wd.shape[3] * groups == Cout
    L = _native.lib()
    dx = torch.empty(x_shape, dtype=F32, device=dy.device)
    M, Ng = B * H * W, Cin // groups
    shape_key = (B, H, W, Cin, Cout) if kh == 1 and groups == 1 else (B, H, W, Cin, Cout, kh, groups)

    def launch(t, dry=False, fuse=_NO_FUSE):
        if dry:
            return _x3_tile_ok(t, Ng)
        _native.check(L.scouter_conv2d_dgrad_x3_bnbwd(_p(dy), _p(wd), _p(addend), _p(dx), B, H, W, Cin, Cout, kh, kh, pad,
                                                      groups, t, *fuse, _stream()), "conv2d_dgrad_x3")
        return True
    fused = _fuse_wanted(post, kh) and post.x_io() == 0
    if fused:
        def launch_fused(t, dry=False):
            if dry:
                return launch(t, dry=True)
            if not post.applied:
                post.alloc((M + 63) // 64, x_shape)
            return launch(t, fuse=post.args())
        if tile is None:
            tile = _pick_tile(("xdgrad+bn", len(post.entries), addend is not None, 3) + shape_key, launch_fused, _X3_TILES)
        tile = L.scouter_conv2d_x3_tile(M, Ng, tile)
        post.alloc(L.scouter_conv2d_x3_halo_partial_rows(groups) if tile == 7 else L.scouter_conv2d_x3_partial_rows(M, Ng, tile),
                   x_shape)
        launch(tile, fuse=post.args())
    else:
        if tile is None:
            tile = _pick_tile(("xdgrad", 3) + shape_key, launch, _X3_TILES)
        launch(L.scouter_conv2d_x3_tile(M, Ng, tile))
    return dx


_X3_WGRAD_PLANS = (-1, 0, 1, 2, 3)


def x3_wgrad_ok(cin, cout):
    return cin % 128 == 0 and cout % 128 == 0


def conv2d_wgrad_x3(x, dy, dw_hwio, plan=None):
    """Weight gradient of a pointwise layer on the register-split bf16x3 GEMM: x [B, H, W, Cin], dy [B, H, W, Cout] fp32 ->
    dw (HWIO [1, 1, Cin, Cout], written in place).  The split-K plan comes from the static table (plans sum the pixels in
    different orders; every plan is deterministic)."""
    _chk(x, "x"); _chk(dy, "dy"); _chk(dw_hwio, "dw")
    B, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    assert tuple(dw_hwio.shape) == (1, 1, Cin, Cout) and tuple(dy.shape[:3]) == (B, H, W)
    L = _native.lib()

    def launch(pl, dry=False):
        if dry:
            return True
        ws = workspace(L.scouter_conv2d_wgrad_x3_workspace_bytes(B, H, W, Cin, Cout, pl), x.device)
        _native.check(L.scouter_conv2d_wgrad_x3(_p(x), _p(dy), _p(dw_hwio), B, H, W, Cin, Cout, pl, _p(ws), ws.numel(),
                                                _stream()), "conv2d_wgrad_x3")
        return True
    if plan is None:
        plan = _pick_tile(("xwgrad", 3, B, H, W, Cin, Cout), launch, _X3_WGRAD_PLANS)
    launch(plan)
    return dw_hwio


def conv2d_wgrad_planes(xp, dyp, dw_hwio, pad, groups=1):
    """xp [np, B, H, W, Cin], dyp [np, B, H, W, Cout] -> dw (HWIO fp32, written in place).  The (tile, split-K) plan is
    autotuned once per layer shape like the fp32 kernel's (_WGRAD_PLANS)."""
    nplanes, B, H, W, Cin = xp.shape
    kh, kw, cg, Cout = dw_hwio.shape
    L = _native.lib()

    def launch(plan, dry=False):
        if dry:         # tap-fused plans (bit 6): 3x3 / pad 1 on maps up to 63 wide, 31-bit byte offsets inside a plane
            return plan < 64 or (kh == 3 and kw == 3 and pad == 1 and W <= 63 and
                                 2 * B * H * W * max(Cin, Cout) < (1 << 31))
        ws = workspace(L.scouter_conv2d_wgrad_planes_workspace_bytes(B, H, W, Cin, Cout, kh, kw, groups, plan), xp.device)
        arr = arrival_counters(xp.device)
        _native.check(L.scouter_conv2d_wgrad_planes(_p(xp), _p(dyp), _p(dw_hwio), B, H, W, Cin, Cout, kh, kw, pad, groups,
                                                    nplanes, plan, _p(ws), ws.numel(), _p(arr), ARRIVAL_SLOTS if arr is not None else 0,
                                                    _stream()), "conv2d_wgrad_planes")
        return True

    launch(_pick_tile(("pwgrad", nplanes, B, H, W, Cin, Cout, kh, kw, pad, groups), launch, _PWGRAD_PLANS))
    return dw_hwio


_side = {}
_side_rr = [0]
SIDE_STREAMS = max(1, int(os.environ.get("SCOUTER_SIDE_STREAMS", "1")))      # weight-gradient side streams (round-robin)
# default of nn_hip.Conv2d.use_side_stream (a per-layer / per-model setting: SlotModel.set_side_stream)
SIDE_STREAM_DEFAULT = os.environ.get("SCOUTER_SIDE_STREAM", "1") != "0"


def _side_priority(which):
    """HIP queue priority of a side stream (torch: -1 high, 0 normal): SCOUTER_SIDE_PRIORITY for the weight-gradient streams,
    SCOUTER_BRANCH_PRIORITY for the shortcut-branch stream.  Development switches (tools_dev/priority_ab.sh)."""
    name = "SCOUTER_BRANCH_PRIORITY" if which == "branch" else "SCOUTER_SIDE_PRIORITY"
    return int(os.environ.get(name, "0"))


class side_stream:
    """Context manager: run the enclosed launches on the per-device WEIGHT-GRADIENT side stream, ordered after
    everything queued so far on the current stream.  In the backward pass only dgrad feeds the next layer; wgrad
    (MFMA-bound) is off the critical path and overlaps the HBM-bound BatchNorm-backward passes of earlier layers.
    `join_side_stream()` makes the current stream wait for it again."""

    def __init__(self, device, *tensors, enabled=True, which="wgrad"):
        """which: "wgrad" (weight gradients) or "branch" (the shortcut branch of a downsampling block, forward and
        backward) -- separate streams, so joining the branch never waits for a backlog of weight gradients."""
        self.device, self.tensors, self.enabled, self.which = device, tensors, enabled, which

    def __enter__(self):
        self.ctx = None
        if not self.enabled:
            return None
        which = self.which
        if which == "wgrad" and SIDE_STREAMS > 1:
            # weight gradients alternate between SIDE_STREAMS streams: the stem's long weight gradients (0.2-0.5 ms each,
            # at the very end of the backward) no longer queue behind one another while the compute stream runs the
            # HBM-bound BatchNorm passes they could overlap with -- join_side_stream waits for all of them
            n = _side_rr[0] = (_side_rr[0] + 1) % SIDE_STREAMS
            which = "wgrad" if n == 0 else "wgrad%d" % n
        key = (self.device.type, self.device.index, which)
        st = _side.get(key)
        if st is None:
            st = _side[key] = torch.cuda.Stream(device=self.device, priority=_side_priority(which))
        st.wait_stream(torch.cuda.current_stream(self.device))
        for t in self.tensors:                 # the caching allocator must not recycle them while the side stream runs
            if t is not None:
                t.record_stream(st)
        self.ctx = torch.cuda.stream(st)
        self.ctx.__enter__()
        return st

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc) if self.ctx is not None else False


def join_side_stream(device, which="wgrad"):
    names = [which] + (["wgrad%d" % n for n in range(1, SIDE_STREAMS)] if which == "wgrad" else [])
    for name in names:
        st = _side.get((device.type, device.index, name))
        if st is not None:
            torch.cuda.current_stream(device).wait_stream(st)


# (tile, split-K) plans of the fp32 / bf16-input weight-gradient kernels: -1 = the library's static plan; the others
# (include/scouter_hip.h plan_hint) are timed once per layer shape like the block tiles.  Every plan is deterministic,
# but different plans sum the pixels in a different order, so -- like plane tile 5 -- the choice is reproducible per
# process (the cache), not across processes; SCOUTER_WGRAD_TUNE=0 or SCOUTER_AUTOTUNE=0 pin the static plan.
# Measured on the benchmark step: +0.7 % images/sec (3 x 120 steps, interleaved, one box).
_WGRAD_PLANS = (-1,)
# SCOUTER_XWT=0: the 32-channel-group 3x3 weight gradients stay on the exact-fp32 MFMA kernels (read by the library too)
XWT = os.environ.get("SCOUTER_XWT", "1") != "0"
# SCOUTER_BHALO=0: the input gradients of those layers in bf16 mode stay on the 128 x 32 tile kernel (read by the library too)
BHALO = os.environ.get("SCOUTER_BHALO", "1") != "0"
# SCOUTER_BWT=0: the bf16-stored twins of those layers stay on the per-tap bf16 kernel (read by the library too)
BWT = os.environ.get("SCOUTER_BWT", "1") != "0"
if os.environ.get("SCOUTER_WGRAD_TUNE", "1") == "1":
    _WGRAD_PLANS += tuple(t | b for t in (0, 16, 32, 48) for b in (0, 1, 2, 3))


# plane weight gradient: the same plans + the TAP-FUSED kernel (csrc/conv_planes_wgrad_taps.h: one workgroup carries all
# nine taps of a 64 x 64 tile from an LDS-resident ring of X rows), bit 6, with a budget of 256 / 512 / 1024 / 2048 workgroups
_PWGRAD_PLANS = _WGRAD_PLANS + ((64, 65, 66, 67) if len(_WGRAD_PLANS) > 1 else ())


def conv2d_wgrad(x, dy, dw_hwio, stride=1, pad=0, groups=1, precision="fp32"):
    """Writes dW (HWIO, contiguous, e.g. a slice of the flat gradient arena).  The (tile, split-K) plan is autotuned
    once per layer shape and kept for the run (see _WGRAD_PLANS)."""
    _chk(x, "x", (F32, BF16)); _chk(dy, "dy", (F32, BF16)); _chk(dw_hwio, "dw")
    B, H, W, Cin = x.shape
    kh, kw, cg, Cout = dw_hwio.shape
    L = _native.lib()
    st = _stream()

    # bf16 matrix inputs where the bf16 kernel applies (same-size stride-1 convolutions, 32-multiples of channels per
    # group -- 32-channel groups on ragged 64-wide tiles); the remaining layers (strided convolutions) keep the fp32 kernel
    same = stride == 1 and dy.shape[1] == H and dy.shape[2] == W
    bf16 = (precision == "bf16" and same and cg % 32 == 0 and (Cout // groups) % 32 == 0 and
            B * H * W >= BF16_MIN_PIXELS and ((kh == 1 and kw == 1 and pad == 0) or 64 // W + 1 < H))
    if (x.dtype != F32 or dy.dtype != F32) and not bf16:
        raise RuntimeError("scouter_amd: a bf16-stored activation / gradient needs the bf16-input weight-gradient kernel "
                           "(precision 'bf16', same-size stride-1 layer); got %s" % (tuple(x.shape),))

    def launch(plan, dry=False):
        if dry:
            return True
        need = L.scouter_conv2d_wgrad_workspace_bytes(B, H, W, Cin, Cout, kh, kw, stride, pad, groups, plan)
        ws = workspace(need, x.device)
        arr = arrival_counters(x.device)
        args = (_p(x), _p(dy), _p(dw_hwio), B, H, W, Cin, Cout, kh, kw, stride, pad, groups, plan, _p(ws), ws.numel(),
                _p(arr), ARRIVAL_SLOTS if arr is not None else 0)
        if bf16:
            _native.check(L.scouter_conv2d_wgrad_bf16_io(
                *args, (IO_X_BF16 if x.dtype == BF16 else 0) | (IO_R_BF16 if dy.dtype == BF16 else 0), st), "conv2d_wgrad")
        else:
            _native.check(L.scouter_conv2d_wgrad_f32(*args, st), "conv2d_wgrad")
        return True

    plan = _pick_tile(("wgrad", bf16, B, H, W, Cin, Cout, kh, kw, stride, pad, groups), launch, _WGRAD_PLANS)
    if XWT and not bf16 and kh == 3 and kw == 3 and stride == 1 and pad == 1 and cg == 32 and W <= 112 and H >= 2:
        # the 32-channel-group 3x3 layers (deep stem, layer1's radix convolutions): the library's own plan, which is the tap-fused
        # register-split bf16x3 kernel (csrc/conv_wgrad_taps_x3.h) -- a static rule of the shape that overrides the table's
        # (tile, split-K) entry for the exact-fp32 kernels, like SCOUTER_X3 does for the layers it moves
        plan = -1
    if (BWT and bf16 and kh == 3 and kw == 3 and stride == 1 and pad == 1 and cg == 32 and W <= 112 and H >= 2 and
            B * H * W * max(Cin, Cout) * 2 < (1 << 31) - (1 << 20)):
        # the same layers in bf16 mode (BASELINE configs[4]): the library's own plan for bf16-STORED operands is the tap-fused
        # one-plane kernel (csrc/conv_wgrad_taps_bf16.h) instead of one workgroup per tap.  The rule is a function of the SHAPE
        # only: an operand stored as fp32 (SCOUTER_BF16_STORAGE=0 / SCOUTER_BF16_GRADS=0) is rounded to bf16 (RNE: the value
        # the per-tap kernel's loader rounds it to) by one extra pass first, so storage stays bit-neutral
        if x.dtype != BF16:
            x = planes_split(x, 1)[0]
        if dy.dtype != BF16:
            dy = planes_split(dy, 1)[0]
        plan = -1
    launch(plan)
    return dw_hwio


def matmul_tn(a, b, out):
    """out[Ka][Kb] = a^T b for row-major a [M][Ka], b [M][Kb] (Ka, Kb multiples of 32): the weight-gradient GEMM of
    a Linear layer, run as a 1x1-conv wgrad (deterministic split over M)."""
    M, Ka = a.shape
    Kb = b.shape[1]
    return conv2d_wgrad(a.view(1, 1, M, Ka), b.view(1, 1, M, Kb), out.view(1, 1, Ka, Kb))


def im2col_nchw(x, k, stride, pad, kpad):
    _chk(x, "image")
    B, Cin, H, W = x.shape
    Ho, Wo = conv_out(H, k, stride, pad), conv_out(W, k, stride, pad)
    col = torch.empty((B, Ho, Wo, kpad), dtype=F32, device=x.device)
    _native.check(_native.lib().scouter_im2col_nchw_f32(_p(x), _p(col), B, Cin, H, W, k, stride, pad, kpad, _stream()),
                  "im2col")
    return col


# SCOUTER_STEM_DIRECT=1: the deep stem's 3 -> 32 convolution (fp32 mode) as ONE direct pass over the NCHW image instead of im2col +
# GEMM (csrc/misc_ops.hip stem_direct_kernel: 88 -> 55 us at 70 x 224^2, no 112 MB of patch rows kept for the backward).  A
# forward-changing option -- three nine-term fmaf chains instead of the MFMA's K order: half the GEMM route's error against
# fp64, other bits -- and like every such option it is one more draw of rounding noise for the single-draw gradient bounds of
# tests/test_model_gpu.py (resnest26d_96: one squeeze-path tensor at 1.06e-3 of its scale against the 1e-3 bound); the step
# does not notice the 34 us (4 927 vs 4 929 img/s, 3 x 60 steps interleaved).  Off by default.
STEM_DIRECT = os.environ.get("SCOUTER_STEM_DIRECT", "0") == "1"


def stem_direct_eligible(Cin, Cout, k, stride, pad, W):
    """The direct kernel of the deep stem's first convolution (csrc/misc_ops.hip stem_direct_kernel): 3 -> 32 channels,
    3x3 / stride 2 / pad 1, rows narrow enough for its LDS stage.  A static rule of the layer shape."""
    return Cin == 3 and Cout == 32 and k == 3 and stride == 2 and pad == 1 and W <= 1000


def stem_direct_fwd(x, w_hwio, bn_stats=False):
    """y[B][Ho][Wo][32] = conv3x3 / 2 of the NCHW image x with w (HWIO); bn_stats: also the (partials, rows) pair conv2d_fwd
    returns for the BatchNorm behind it."""
    _chk(x, "image"); _chk(w_hwio, "weight")
    B, Cin, H, W = x.shape
    Cout = w_hwio.shape[-1]
    Ho, Wo = conv_out(H, 3, 2, 1), conv_out(W, 3, 2, 1)
    L = _native.lib()
    y = torch.empty((B, Ho, Wo, Cout), dtype=F32, device=x.device)
    part, rows = None, 0
    if bn_stats:
        rows = L.scouter_stem_direct_partial_rows(B, H)
        part = torch.empty((rows, Cout, 2), dtype=torch.float64, device=x.device)
    _native.check(L.scouter_stem_direct_fwd_f32(_p(x), _p(w_hwio), _p(y), _p(part), B, H, W, Cout, _stream()), "stem_direct_fwd")
    return (y, (part, rows)) if bn_stats else y


def pad_rows(w_flat, nvalid, ntotal):
    out = torch.empty(ntotal, dtype=F32, device=w_flat.device)
    _native.check(_native.lib().scouter_pad_rows_f32(_p(w_flat), _p(out), nvalid, ntotal, _stream()), "pad_rows")
    return out


# ---------------------------------------------------------------------------------------------------------------
# batch norm / elementwise
# ---------------------------------------------------------------------------------------------------------------
def _col_ws(M, C, device):
    L = _native.lib()
    return workspace(L.scouter_colreduce_workspace_bytes(M, C) + 8 * C + 64, device)


def bn_stats(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, stats=None):
    """Finalises the BatchNorm statistics only (running stats updated in training): returns saved = [4, C] (mean, rstd,
    scale, shift).  The consumer applies (x - mean) * scale + shift itself (fused split attention)."""
    _chk(x, "x", (F32, BF16))      # (a bf16-stored x is never read: its statistics are `stats` / the running ones)
    C = x.shape[-1]
    M = x.numel() // C
    saved = torch.empty((4, C), dtype=F32, device=x.device)
    ws = _col_ws(M, C, x.device)
    _native.check(_native.lib().scouter_bn_fwd_io(
        _p(x), None, None, M, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), momentum, eps,
        int(training), 0, _p(saved[0]), _p(saved[1]), _p(saved[2]), _p(saved[3]),
        _p(stats[0]) if stats else None, stats[1] if stats else 0, None, None, 0, None,
        IO_X_BF16 if x.dtype == BF16 else 0, _p(ws), ws.numel(), _stream()), "bn_stats")
    return saved


def bn_apply(x, saved, relu):
    """y = [relu]((x - mean) * scale + shift) with a `saved` block from bn_stats / bn_fwd: the apply pass alone (the same
    kernel and expression the fused consumers evaluate; used by the tests to read the ReLU sign pattern)."""
    y = torch.empty_like(x)
    C = x.shape[-1]
    M = x.numel() // C
    _native.check(_native.lib().scouter_bn_apply_f32(_p(x), _p(saved), _p(y), M, C, int(relu), _stream()), "bn_apply")
    return y


def bn_fwd(x, gamma, beta, running_mean, running_var, training, relu, residual=None, momentum=0.1, eps=1e-5,
           stats=None, want_mask=False, planes=0, residual_bn=None, keep_f32=True, out_dtype=F32):
    """x: [..., C] NHWC.  Returns (y, saved) with saved = (mean, rstd, scale, shift) packed as one [4, C] tensor.
    stats = (partial, rows) from conv2d_fwd(bn_stats=True) replaces the statistics pass over x.
    want_mask (with relu): returns (y, saved, mask) -- the 1-bit/element sign mask bn_bwd takes instead of y.
    x / residual may be bfloat16-stored tensors and out_dtype=torch.bfloat16 stores y rounded (activation storage)."""
    _chk(x, "x", (F32, BF16)); _chk(residual, "residual", (F32, BF16))
    C = x.shape[-1]
    M = x.numel() // C
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device) if keep_f32 or not planes else None   # keep_f32=False: planes only
    io = ((IO_X_BF16 if x.dtype == BF16 else 0) | (IO_Y_BF16 if out_dtype == BF16 else 0) |
          (IO_R_BF16 if residual is not None and residual.dtype == BF16 else 0))
    saved = torch.empty((4, C), dtype=F32, device=x.device)
    ws = _col_ws(M, C, x.device)
    mask = None
    if want_mask:
        assert relu
        mask = torch.empty(_native.lib().scouter_relu_mask_words(x.numel()), dtype=torch.int64, device=x.device)
    yp = torch.empty((planes,) + tuple(x.shape), dtype=BF16, device=x.device) if planes else None
    _native.check(_native.lib().scouter_bn_fwd_io(
        _p(x), _p(y), _p(residual), M, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), momentum, eps,
        int(training), int(relu), _p(saved[0]), _p(saved[1]), _p(saved[2]), _p(saved[3]),
        _p(stats[0]) if stats else None, stats[1] if stats else 0, _p(mask), _p(yp), planes, _p(residual_bn), io, _p(ws),
        ws.numel(), _stream()), "bn_fwd")
    if planes:
        y = PlaneTensor(y, yp)
    return (y, saved, mask) if want_mask else (y, saved)


def bn_bwd(dy, ymask, x, saved, training, dgamma=None, dbeta=None, want_gout=False, mask=None, ext=None, dx_dtype=F32):
    """ReLU sign from `mask` (bits written by bn_fwd(want_mask=True)) or from the activation `ymask`; both None: no ReLU.
    ext = (partial, rows) from a BnBwdFuse: dy is the already masked gradient, its sums are reduced -- no reduction
    pass, no mask, and the masked gradient `gout` is dy itself.  dx_dtype=torch.bfloat16: dx is stored as bf16 (for a dx
    read only by bf16-input convolution kernels, which round it the same way)."""
    _chk(dy, "dy", (F32, BF16)); _chk(ymask, "ymask"); _chk(x, "x", (F32, BF16))
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty(x.shape, dtype=dx_dtype, device=x.device)     # (x may be bf16-stored)
    ws = _col_ws(M, C, x.device)
    io = ((IO_X_BF16 if x.dtype == BF16 else 0) | (IO_Y_BF16 if dx_dtype == BF16 else 0) |
          (IO_R_BF16 if dy.dtype == BF16 else 0))
    if dy.dtype == BF16 and ext is None and want_gout:
        raise RuntimeError("scouter_amd: a bf16-stored dy is the already masked gradient (its producer fused the mask)")
    if ext is not None:
        part, rows = ext
        _native.check(_native.lib().scouter_bn_bwd_io(
            _p(dy), None, _p(x), _p(saved[0]), _p(saved[1]), _p(saved[2]), None, M, C, int(training), _p(dgamma),
            _p(dbeta), _p(dx), None, _p(part), rows, io, _p(ws), ws.numel(), _stream()), "bn_bwd")
        return dx, (dy if want_gout else None)
    gout = torch.empty(x.shape, dtype=F32, device=x.device) if want_gout else None
    _native.check(_native.lib().scouter_bn_bwd_io(
        _p(dy), _p(ymask), _p(x), _p(saved[0]), _p(saved[1]), _p(saved[2]), _p(mask), M, C, int(training), _p(dgamma),
        _p(dbeta), _p(dx), _p(gout), None, 0, io, _p(ws), ws.numel(), _stream()), "bn_bwd")
    return dx, gout


def colsum(a, out, b=None, alpha=1.0):
    C = a.shape[-1]
    M = a.numel() // C
    ws = _col_ws(M, C, a.device)
    _native.check(_native.lib().scouter_colsum_f32(_p(a), _p(b), _p(out), M, C, alpha, _p(ws), ws.numel(), _stream()),
                  "colsum")
    return out


def iadd_i64(t, v=1):
    """t (contiguous int64) += v, one launch."""
    assert t.dtype == torch.int64 and t.is_contiguous() and t.is_cuda
    _native.check(_native.lib().scouter_iadd_i64(_p(t), t.numel(), int(v), _stream()), "iadd_i64")
    return t


def relu_bwd(dy, y):
    dx = torch.empty_like(dy)
    _native.check(_native.lib().scouter_relu_bwd_f32(_p(dy), _p(y), _p(dx), dy.numel(), _stream()), "relu_bwd")
    return dx


def axpby(a, b=None, alpha=1.0, beta=1.0, out=None):
    out = torch.empty_like(a) if out is None else out
    _native.check(_native.lib().scouter_axpby_f32(_p(a), _p(b), _p(out), alpha, beta, a.numel(), _stream()), "axpby")
    return out


# ---------------------------------------------------------------------------------------------------------------
# pooling / layout
# ---------------------------------------------------------------------------------------------------------------
def pool_out(n, k, s, p, ceil_mode=False):
    return _native.lib().scouter_pool_out_size(n, k, s, p, int(ceil_mode))


def maxpool_fwd(x, k=3, stride=2, pad=1, want_argmax=True):
    _chk(x, "x")
    B, H, W, C = x.shape
    y = torch.empty((B, pool_out(H, k, stride, pad), pool_out(W, k, stride, pad), C), dtype=F32, device=x.device)
    arg = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if want_argmax else None
    _native.check(_native.lib().scouter_maxpool_fwd_f32(_p(x), _p(y), _p(arg), B, H, W, C, k, stride, pad, _stream()),
                  "maxpool_fwd")
    return y, arg


def maxpool_bwd(dy, arg, x_shape, k=3, stride=2, pad=1):
    B, H, W, C = x_shape
    dx = torch.empty(x_shape, dtype=F32, device=dy.device)
    _native.check(_native.lib().scouter_maxpool_bwd_f32(_p(dy), _p(arg), _p(dx), B, H, W, C, k, stride, pad, _stream()),
                  "maxpool_bwd")
    return dx


def bn_maxpool_fwd(x, saved, k=3, stride=2, pad=1, want_argmax=True):
    """maxpool(relu(bn(x))) with the BatchNorm applied on the fly from its saved block (bn_stats): the activation is
    never stored.  Returns (pooled, argmax taps)."""
    _chk(x, "x"); _chk(saved, "saved")
    B, H, W, C = x.shape
    y = torch.empty((B, pool_out(H, k, stride, pad), pool_out(W, k, stride, pad), C), dtype=F32, device=x.device)
    arg = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if want_argmax else None
    _native.check(_native.lib().scouter_bn_maxpool_fwd_f32(_p(x), _p(saved), _p(y), _p(arg), B, H, W, C, k, stride, pad,
                                                           _stream()), "bn_maxpool_fwd")
    return y, arg


def bn_maxpool_bwd(dy, arg, x, saved, training, dgamma=None, dbeta=None, k=3, stride=2, pad=1, dx_dtype=F32):
    """Backward of bn_maxpool_fwd: pooled gradient -> gradient of the BatchNorm input (+ dgamma, dbeta in place).
    dx_dtype=torch.bfloat16: dx stored as bf16 (read by bf16-input convolution kernels only: Conv2d.grad_storage)."""
    _chk(dy, "dy"); _chk(x, "x")
    B, H, W, C = x.shape
    dx = torch.empty(x.shape, dtype=dx_dtype, device=x.device)
    ws = _col_ws(dy.numel() // C, C, x.device)
    _native.check(_native.lib().scouter_bn_maxpool_bwd_io(
        _p(dy), _p(arg), _p(x), _p(saved), B, H, W, C, k, stride, pad, int(training), _p(dgamma), _p(dbeta), _p(dx),
        IO_Y_BF16 if dx_dtype == BF16 else 0, _p(ws), ws.numel(), _stream()), "bn_maxpool_bwd")
    return dx


def avgpool_fwd(x, k, stride, pad, ceil_mode, count_include_pad):
    """x may be a bfloat16-stored activation; the pooled tensor is fp32."""
    _chk(x, "x", (F32, BF16))
    B, H, W, C = x.shape
    y = torch.empty((B, pool_out(H, k, stride, pad, ceil_mode), pool_out(W, k, stride, pad, ceil_mode), C), dtype=F32,
                    device=x.device)
    _native.check(_native.lib().scouter_avgpool_fwd_io(_p(x), _p(y), B, H, W, C, k, stride, pad, int(ceil_mode),
                                                       int(count_include_pad), IO_X_BF16 if x.dtype == BF16 else 0,
                                                       _stream()), "avgpool_fwd")
    return y


def avgpool_bwd(dy, x_shape, k, stride, pad, ceil_mode, count_include_pad):
    B, H, W, C = x_shape
    dx = torch.empty(x_shape, dtype=F32, device=dy.device)
    _native.check(_native.lib().scouter_avgpool_bwd_f32(_p(dy), _p(dx), B, H, W, C, k, stride, pad, int(ceil_mode),
                                                        int(count_include_pad), _stream()), "avgpool_bwd")
    return dx


def nchw_to_nhwc(x):
    _chk(x, "x")
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=F32, device=x.device)
    _native.check(_native.lib().scouter_transpose_f32(_p(x), _p(out), B, C, H * W, _stream()), "transpose")
    return out


def nhwc_to_nchw(x):
    _chk(x, "x")
    B, H, W, C = x.shape
    out = torch.empty((B, C, H, W), dtype=F32, device=x.device)
    _native.check(_native.lib().scouter_transpose_f32(_p(x), _p(out), B, H * W, C, _stream()), "transpose")
    return out


# ---------------------------------------------------------------------------------------------------------------
# split attention glue
# ---------------------------------------------------------------------------------------------------------------
def _sa_ws(B, HW, C2, device):
    return workspace(_native.lib().scouter_sa_workspace_bytes(B, HW, C2), device)


def sa_gap(x, bn=None):
    """x: [B, H, W, 2C'] -> gap [B, C'] = mean_hw(h[..., :C'] + h[..., C':]), h = x, or relu(bn(x)) on the fly when `bn`
    (the [4, 2C'] block of bn_stats) is given and x is the raw convolution output."""
    _chk(x, "x", (F32, BF16))
    B, H, W, C2 = x.shape
    out = torch.empty((B, C2 // 2), dtype=F32, device=x.device)
    ws = _sa_ws(B, H * W, C2, x.device)
    _native.check(_native.lib().scouter_sa_reduce_io(_p(x), None, _p(bn), _p(out), None, B, H * W, C2 // 2, 0,
                                                     IO_X_BF16 if x.dtype == BF16 else 0, _p(ws), ws.numel(), _stream()),
                  "sa_gap")
    return out


def sa_dattn(x, dout, bn=None, want_stats=False):
    """da[b, r*C'+c] = sum_hw dout[b,hw,c] * h[b,hw,r*C'+c]   (h as in sa_gap).  want_stats (needs bn): returns
    (da, sums) with the per-image statistics [B, 2C', 4] (fp64) that let sa_bn_bwd skip its reduction pass."""
    _chk(x, "x", (F32, BF16)); _chk(dout, "dout")
    B, H, W, C2 = x.shape
    out = torch.empty((B, C2), dtype=F32, device=x.device)
    sums = torch.empty((B, C2, 4), dtype=torch.float64, device=x.device) if want_stats else None
    ws = _sa_ws(B, H * W, C2, x.device)
    _native.check(_native.lib().scouter_sa_reduce_io(_p(x), _p(dout), _p(bn), _p(out), _p(sums), B, H * W, C2 // 2, 1,
                                                     IO_X_BF16 if x.dtype == BF16 else 0, _p(ws), ws.numel(), _stream()),
                  "sa_dattn")
    return (out, sums) if want_stats else out


def sa_bn_bwd(dout, a, dgap, x0, bn, training, dgamma=None, dbeta=None, planes=0, keep_f32=True, sums=None):
    """Backward of [bn0 -> ReLU -> split-attention weighting / GAP] in one fused chain: returns the gradient w.r.t. the
    radix convolution's raw output x0 [B, H, W, 2C'] (a PlaneTensor when `planes`; keep_f32=False: planes only) and fills
    dgamma / dbeta."""
    _chk(x0, "x0", (F32, BF16)); _chk(dout, "dout")
    B, H, W, Cp = dout.shape
    dx = torch.empty(x0.shape, dtype=F32, device=x0.device) if keep_f32 or not planes else None
    dxp = torch.empty((planes,) + tuple(x0.shape), dtype=BF16, device=x0.device) if planes else None
    ws = _col_ws(B * H * W, 2 * Cp, x0.device)
    _native.check(_native.lib().scouter_sa_bn_bwd_io(_p(dout), _p(a), _p(dgap), _p(x0), _p(bn), _p(sums), B, H * W, Cp,
                                                     int(training), _p(dgamma), _p(dbeta), _p(dx), _p(dxp), planes,
                                                     IO_X_BF16 if x0.dtype == BF16 else 0, _p(ws), ws.numel(), _stream()),
                  "sa_bn_bwd")
    return PlaneTensor(dx, dxp) if planes else dx


def radix_softmax_fwd(z):
    a = torch.empty_like(z)
    _native.check(_native.lib().scouter_radix_softmax_fwd_f32(_p(z), _p(a), z.shape[0], z.shape[1] // 2, _stream()),
                  "radix_softmax_fwd")
    return a


def radix_softmax_bwd(a, da):
    dz = torch.empty_like(a)
    _native.check(_native.lib().scouter_radix_softmax_bwd_f32(_p(a), _p(da), _p(dz), a.shape[0], a.shape[1] // 2,
                                                              _stream()), "radix_softmax_bwd")
    return dz


def sa_apply_fwd(x, a, bn=None, out_dtype=F32):
    """x may be bf16-stored; out_dtype=torch.bfloat16 stores the weighted sum as bf16."""
    _chk(x, "x", (F32, BF16))
    B, H, W, C2 = x.shape
    out = torch.empty((B, H, W, C2 // 2), dtype=out_dtype, device=x.device)
    _native.check(_native.lib().scouter_sa_apply_fwd_io(
        _p(x), _p(a), _p(bn), _p(out), B, H * W, C2 // 2,
        (IO_X_BF16 if x.dtype == BF16 else 0) | (IO_Y_BF16 if out_dtype == BF16 else 0), _stream()), "sa_apply_fwd")
    return out


def sa_apply_bwd(dout, a, dgap):
    B, H, W, Cp = dout.shape
    dx = torch.empty((B, H, W, 2 * Cp), dtype=F32, device=dout.device)
    _native.check(_native.lib().scouter_sa_apply_bwd_f32(_p(dout), _p(a), _p(dgap), _p(dx), B, H * W, Cp, _stream()),
                  "sa_apply_bwd")
    return dx


# ---------------------------------------------------------------------------------------------------------------
# xSlot head
# ---------------------------------------------------------------------------------------------------------------
_pe_cache = {}


def posenc_sine(h, w, d, device):
    """Token-major sine positional encoding [h*w, d] (constant per grid; cached per device)."""
    key = (h, w, d, device.type, device.index)
    pe = _pe_cache.get(key)
    if pe is None:
        pe = torch.empty((h * w, d), dtype=F32, device=device)
        _native.check(_native.lib().scouter_posenc_sine_f32(_p(pe), h, w, d, _stream()), "posenc_sine")
        _pe_cache[key] = pe
    return pe


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def xslot_fwd(X, PE, tok_w, tok_b, slots0, w_ih, w_hh, b_ih, b_hh, spc, T, loss_status):
    """tok_w / tok_b: lists of the L to_k Linear weights [d,d] (out,in) / biases [d]."""
    for n, t in (("X", X), ("PE", PE), ("slots0", slots0), ("w_ih", w_ih), ("w_hh", w_hh), ("b_ih", b_ih),
                 ("b_hh", b_hh)) + tuple(("to_k", t) for t in list(tok_w) + list(tok_b)):
        _chk(t, n)
    B, N, d = X.shape
    S, L = slots0.shape[0], len(tok_w)
    dev = X.device
    out = dict(
        logits=torch.empty((B, S // spc), dtype=F32, device=dev), attn=torch.empty((B, S, N), dtype=F32, device=dev),
        area_part=torch.empty((B,), dtype=F32, device=dev), K=torch.empty((B, N, d), dtype=F32, device=dev),
        H=torch.empty((L, B, N, d), dtype=F32, device=dev),
        states=torch.empty((max(T - 1, 1), B, S, d), dtype=F32, device=dev))
    _native.check(_native.lib().scouter_xslot_fwd_f32(
        _p(X), _p(PE), _ptr_array(tok_w), _ptr_array(tok_b), _p(slots0), _p(w_ih), _p(w_hh), _p(b_ih), _p(b_hh), B, N, d,
        S, spc, T, L,
        float(loss_status), _p(out["logits"]), _p(out["attn"]), _p(out["area_part"]), _p(out["K"]), _p(out["H"]),
        _p(out["states"]), _stream()), "xslot_fwd")
    return out


def xslot_bwd(X, PE, tok_w, slots0, w_ih, w_hh, b_ih, b_hh, saved, dlogits, g_area_sum, spc, T, loss_status):
    B, N, d = X.shape
    S, L = slots0.shape[0], len(tok_w)
    dev = X.device
    Lb = _native.lib()
    ws = workspace(Lb.scouter_xslot_bwd_workspace_bytes(B, N, d, S, T), dev)
    nt = max(T - 1, 1)
    out = dict(dX=torch.empty((B, N, d), dtype=F32, device=dev),
               dgi=torch.empty((nt, B, S, 3 * d), dtype=F32, device=dev),
               dgh=torch.empty((nt, B, S, 3 * d), dtype=F32, device=dev),
               U=torch.empty((nt, B, S, d), dtype=F32, device=dev),
               ds0=torch.empty((B, S, d), dtype=F32, device=dev), dZ=torch.empty((L, B, N, d), dtype=F32, device=dev))
    _native.check(Lb.scouter_xslot_bwd_f32(
        _p(X), _p(PE), _ptr_array(tok_w), _p(slots0), _p(w_ih), _p(w_hh), _p(b_ih), _p(b_hh), _p(saved["K"]),
        _p(saved["H"]),
        _p(saved["states"]), _p(dlogits), _p(g_area_sum), B, N, d, S, spc, T, L, float(loss_status), _p(out["dX"]),
        _p(out["dgi"]), _p(out["dgh"]), _p(out["U"]), _p(out["ds0"]), _p(out["dZ"]), _p(ws), ws.numel(), _stream()),
        "xslot_bwd")
    return out


def slot_loss_fwd(logits, labels, area_part, area_count, lam, power):
    """Returns (log_probs [B,C], stats [5] = loss, nll, area**power, top-1, area)."""
    B, C = logits.shape
    logp = torch.empty_like(logits)
    stats = torch.empty(8, dtype=F32, device=logits.device)      # (the kernel writes all eight words)
    _chk(labels, "labels", torch.int64)
    _native.check(_native.lib().scouter_slot_loss_fwd_f32(
        _p(logits), _p(labels), _p(area_part), 0 if area_part is None else area_part.numel(), B, C, float(area_count),
        float(lam), float(power), _p(logp), _p(stats), _stream()), "slot_loss_fwd")
    return logp, stats


def slot_loss_bwd(logp, labels, stats, g_loss, g_nll, g_term, g_logp, area_count, lam, power):
    B, C = logp.shape
    dlogits = torch.empty_like(logp)
    g_area = torch.empty(1, dtype=F32, device=logp.device)
    _native.check(_native.lib().scouter_slot_loss_bwd_f32(
        _p(logp), _p(labels), _p(stats), _p(g_loss), _p(g_nll), _p(g_term), _p(g_logp), B, C, float(area_count),
        float(lam), float(power), _p(dlogits), _p(g_area), _stream()), "slot_loss_bwd")
    return dlogits, g_area


# ---------------------------------------------------------------------------------------------------------------
# FC baseline head
# ---------------------------------------------------------------------------------------------------------------
def linear_small_fwd(x, w, bias):
    _chk(x, "x"); _chk(w, "weight"); _chk(bias, "bias")
    B, Kd = x.shape
    C = w.shape[0]
    y = torch.empty((B, C), dtype=F32, device=x.device)
    _native.check(_native.lib().scouter_linear_small_fwd_f32(_p(x), _p(w), _p(bias), _p(y), B, Kd, C, _stream()),
                  "linear_small_fwd")
    return y


def linear_small_bwd(dy, x, w, dw=None, db=None, need_dx=True):
    B, Kd = x.shape
    C = w.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    _native.check(_native.lib().scouter_linear_small_bwd_f32(_p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), B, Kd, C,
                                                             _stream()), "linear_small_bwd")
    return dx


# ---- input pipeline (dataset/transform_func.py:101-124 on the GPU)
def resize_normalize(images_u8, out_size, lut, out=None):
    """images_u8: list of B uint8 device tensors [h, w, C] (any sizes, same C) -> float32 [B, C, S, S]:
    PIL-exact bilinear resize + ToTensor + Normalize (`lut` = [C, 256] float32 table).  Two launches per batch."""
    B = len(images_u8)
    dev = images_u8[0].device
    C = images_u8[0].shape[2]
    for t in images_u8:
        assert t.is_cuda and t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == C and t.is_contiguous(), \
            "resize_normalize: dense uint8 [h, w, C] device tensors expected"
    hs, ws = [int(t.shape[0]) for t in images_u8], [int(t.shape[1]) for t in images_u8]
    offs, tot = [], 0
    for h in hs:
        offs.append(tot)
        tot += (h * out_size * C + 15) // 16 * 16
    meta = torch.tensor([t.data_ptr() for t in images_u8] + offs + [v for hw in zip(hs, ws) for v in hw],
                        dtype=torch.int64)
    meta = meta.to(dev, non_blocking=True)                    # [B ptrs | B offsets | B x (h, w)]
    hw32 = meta[2 * B:].to(torch.int32)
    tmp = workspace(tot + 16, dev)
    if out is None:
        out = torch.empty((B, C, out_size, out_size), dtype=F32, device=dev)
    _native.check(_native.lib().scouter_resize_normalize_u8_f32(
        _p(meta), _p(hw32), _p(tmp), _p(meta[B:]), _p(lut), _p(out), B, C, max(hs), max(ws), out_size, out_size,
        _stream()), "resize_normalize")
    return out
