"""The resident-rows bf16 forward (csrc/conv_halo_dgrad_bf16.h bhalo_fwd_kernel, SCOUTER_BHALO=1) against the 128-pixel tile kernels
on config 5's 32-input-channel-group 3x3 layers at batch B (fused BatchNorm statistics, bf16-stored x): bit identity, microseconds
(library hipEvents).  usage: python tools_dev/bhalo_fwd_check.py [B=256]"""
import ctypes, os, sys
import torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K, _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
L = _native.lib()
buf = ctypes.create_string_buffer(1 << 14)


def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    L.scouter_prof_collect(buf, len(buf)); L.scouter_prof_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); L.scouter_prof_enable(0); L.scouter_prof_collect(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split("\t")[:3]
        out[name] = float(ms) / float(cnt) * 1e3
    return out


for (b, H, W, Cin, Cout, groups) in [(8, 24, 24, 64, 128, 2), (B, 112, 112, 32, 32, 1), (B, 112, 112, 32, 64, 1), (B, 56, 56, 64, 128, 2)]:
    x = r(b, H, W, Cin).to(torch.bfloat16)
    w = r(3, 3, Cin // groups, Cout) * 0.1
    res, tm = {}, {}
    for flag in ("0", "1"):
        os.environ["SCOUTER_BHALO"] = flag
        K.BHALO = flag == "1"
        K._tile_cache[("fwd", True, b, H, W, Cin, Cout, 3, 3, 1, 1, groups)] = 1 if Cout // groups == 64 else 3
        fn = lambda: K.conv2d_fwd(x, w, None, None, 1, 1, groups, False, bn_stats=True, precision="bf16", out_dtype=torch.bfloat16)
        tm[flag] = {k: v for k, v in timed(fn).items() if "weight" not in k}
        y, (p, rows) = fn()
        res[flag] = (y.clone(), p[:rows].clone())
    same = torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])
    print("B=%d %dx%d %d->%d g%d: bit-identical %s | us: " % (b, H, W, Cin, Cout, groups, same)
          + " vs ".join(", ".join("%s %.1f" % kv for kv in tm[f].items()) for f in ("0", "1")), flush=True)
K._tile_cache.clear()
os.environ.pop("SCOUTER_BHALO", None)
