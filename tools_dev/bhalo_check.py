"""The resident-rows bf16 input gradient (csrc/conv_halo_dgrad_bf16.h, SCOUTER_BHALO=1) against the 128 x 32 tile kernel on the
32-input-channel-group 3x3 layers (config 5's shapes at batch B): relative difference, run-to-run identity, microseconds
(library hipEvents).  usage: python tools_dev/bhalo_check.py [B=256]"""
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


shapes = [(8, 24, 24, 64, 128, 2), (B, 112, 112, 32, 32, 1), (B, 112, 112, 32, 64, 1), (B, 56, 56, 64, 128, 2)]
for (b, H, W, Cin, Cout, groups) in shapes:
    dy = r(b, H, W, Cout).to(torch.bfloat16)
    w = r(3, 3, Cin // groups, Cout) * 0.1
    x1 = r(b, H, W, Cin).to(torch.bfloat16)
    gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.zeros(Cin, device=dev)
    _, saved, mask = K.bn_fwd(x1.float(), gamma, beta, torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), True, True,
                              want_mask=True)
    res, tm = {}, {}
    for flag in ("0", "1"):
        os.environ["SCOUTER_BHALO"] = flag
        K.BHALO = flag == "1"
        def fn():
            post = K.BnBwdFuse(mask, [(x1, saved)])
            return K.conv2d_dgrad(dy, w, (b, H, W, Cin), None, 1, 1, groups, precision="bf16", post=post)
        tm[flag] = timed(fn)
        res[flag] = fn().clone()
        torch.cuda.synchronize()
        assert torch.equal(res[flag], fn()), "not reproducible"
    d = float((res["0"] - res["1"]).abs().max()) / float(res["0"].abs().max())
    print("B=%d %dx%d %d<-%d g%d (fused BatchNorm backward, bf16-stored x): rel diff %.2e | us: " % (b, H, W, Cin, Cout, groups, d)
          + " vs ".join(", ".join("%s %.1f" % kv for kv in tm[f].items()) for f in ("0", "1")), flush=True)
os.environ.pop("SCOUTER_BHALO", None)
