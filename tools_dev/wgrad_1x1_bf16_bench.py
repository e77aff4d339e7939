"""1x1 weight gradients of BASELINE configs[4] (batch 256) with BOTH operands bf16-stored: the bf16-input igemm kernel
(register-transposed patches) vs the one-plane plane kernel (LDS-DMA + transposing LDS reads), every plan."""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K

def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tot_i = tot_p = 0.0
for H, cin, cout, n in [(56, 256, 64, 2), (56, 64, 256, 3), (28, 512, 128, 3), (28, 128, 512, 4), (14, 1024, 256, 5), (14, 256, 1024, 6),
                        (7, 2048, 512, 2), (7, 512, 2048, 3), (56, 64, 64, 1), (56, 256, 128, 1), (28, 512, 256, 1), (14, 1024, 512, 1)]:
    x = torch.randn(B, H, H, cin, device='cuda').to(torch.bfloat16); dy = torch.randn(B, H, H, cout, device='cuda').to(torch.bfloat16)
    dw = torch.empty(1, 1, cin, cout, device='cuda')
    ki = ("wgrad", True, B, H, H, cin, cout, 1, 1, 1, 0, 1)
    kp = ("pwgrad", 1, B, H, H, cin, cout, 1, 1, 0, 1)
    bi = bp = None
    for p in K._WGRAD_PLANS:
        K._tile_cache[ki] = p
        t = timeit(lambda: K.conv2d_wgrad(x, dy, dw, 1, 0, 1, precision="bf16"))
        bi = (t, p) if bi is None or t < bi[0] else bi
    for p in K._WGRAD_PLANS:
        K._tile_cache[kp] = p
        try:
            t = timeit(lambda: K.conv2d_wgrad_planes(x.unsqueeze(0), dy.unsqueeze(0), dw, 0, 1))
        except RuntimeError as e:
            continue
        bp = (t, p) if bp is None or t < bp[0] else bp
    gb = 2.0 * B * H * H * (cin + cout) / 1e9
    print("wgrad 1x1 %4d->%4d @%2d x%d: igemm_bf16 %.1f us (plan %d, %.0f GB/s) | one-plane %s" % (
        cin, cout, H, n, bi[0], bi[1], gb / bi[0] * 1e6, "%.1f us (plan %d, %.0f GB/s)" % (bp[0], bp[1], gb / bp[0] * 1e6) if bp else "unsupported"))
    tot_i += n * bi[0]; tot_p += n * (bp[0] if bp else bi[0])
print("per step: igemm_bf16 %.2f ms, one-plane %.2f ms" % (tot_i / 1e3, tot_p / 1e3))
