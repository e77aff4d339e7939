"""The bf16 tap-fused weight gradient (csrc/conv_wgrad_taps_bf16.h, SCOUTER_BWT=1) against the per-tap bf16 kernel and an fp64
reference of the same bf16 values on the 32-channel-group 3x3 layers: error, run-to-run identity, microseconds (library hipEvents).
usage: python tools_dev/bwt_check.py [B=256]"""
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


def ref64(x, dy, groups):
    xc = x.permute(0, 3, 1, 2).double().cpu(); dyc = dy.permute(0, 3, 1, 2).double().cpu()
    w = torch.zeros(dy.shape[-1], x.shape[-1] // groups, 3, 3, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(xc, w, padding=1, groups=groups)
    (y * dyc).sum().backward()
    return w.grad.permute(2, 3, 1, 0).contiguous()          # HWIO


shapes = [(8, 12, 12, 32, 32, 1), (7, 9, 20, 32, 64, 1), (8, 14, 10, 64, 128, 2), (2, 112, 112, 32, 32, 1),
          (B, 112, 112, 32, 32, 1), (B, 112, 112, 32, 64, 1), (B, 56, 56, 64, 128, 2)]
for (b, H, W, Cin, Cout, groups) in shapes:
    x, dy = r(b, H, W, Cin).to(torch.bfloat16), (r(b, H, W, Cout) * 0.1).to(torch.bfloat16)
    res, tm = {}, {}
    for flag in ("0", "1"):
        os.environ["SCOUTER_BWT"] = flag
        K.BWT = flag == "1"
        dw = torch.empty(3, 3, Cin // groups, Cout, device=dev)
        fn = lambda: K.conv2d_wgrad(x, dy, dw, 1, 1, groups, precision="bf16")
        tm[flag] = timed(fn)
        res[flag] = dw.clone()
        fn(); torch.cuda.synchronize()
        assert torch.equal(res[flag], dw), "not reproducible"
    line = "B=%d %dx%d %d->%d g%d: " % (b, H, W, Cin, Cout, groups)
    if b * H * W <= 30000:
        ref = ref64(x, dy, groups)
        e0 = float((res["0"].cpu().double() - ref).abs().max()); e1 = float((res["1"].cpu().double() - ref).abs().max())
        line += "err per-tap %.2e taps %.2e (max|ref| %.2f) " % (e0, e1, float(ref.abs().max()))
    d = float((res["0"] - res["1"]).abs().max()) / float(res["0"].abs().max())
    line += "rel diff %.2e | us: " % d + " vs ".join(", ".join("%s %.1f" % kv for kv in tm[f].items()) for f in ("0", "1"))
    print(line, flush=True)
os.environ.pop("SCOUTER_BWT", None)
