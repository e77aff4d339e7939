import sys, time, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K, _native
cin, cout, k, g, H, B = 128, 256, 3, 2, 56, 70
x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
y = torch.empty(B, H, H, cout, device='cuda')
L = _native.lib(); st = torch.cuda.current_stream().cuda_stream
fl = 2.0 * y.numel() * (cin // g) * k * k
for abl in (0, 1, 2, 4, 3, 6, 7):
    f = lambda: L.scouter_conv2d_fwd_f32(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, H, H, cin, cout, k, k, 1, 1, g, abl << 8, st)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print('ablate %d (1=noglobal 2=nobarrier 4=nolds): %.1f us  %.1f TF' % (abl, t * 1e6, fl / t / 1e12))
