import sys, time, subprocess, os
shapes = [(64, 256, 1, 0, 1, 56), (64, 64, 1, 0, 1, 56), (256, 64, 1, 0, 1, 56), (256, 128, 1, 0, 1, 56), (128, 512, 1, 0, 1, 28), (512, 128, 1, 0, 1, 28), (256, 1024, 1, 0, 1, 14), (512, 2048, 1, 0, 1, 7), (64, 128, 3, 1, 2, 56), (128, 256, 3, 1, 2, 28)]
code = r'''
import sys, time, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
shapes = %r
for (cin, cout, k, p, g, H) in shapes:
    x = torch.randn(70, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
    y = K.conv2d_fwd(x, w, None, None, 1, p, g); dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * (cin // g) * k * k
    res = []
    for fn in (lambda: K.conv2d_fwd(x, w, None, None, 1, p, g), lambda: K.conv2d_dgrad(dy, w, tuple(x.shape), None, 1, p, g)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); res.append(fl / ((time.perf_counter() - t0) / 10) / 1e12)
    print('%%-28s fwd %%6.1f  dgrad %%6.1f' %% (str((cin, cout, k, g, H)), res[0], res[1]))
''' % (shapes,)
for t in ['', '0', '1', '2', '3']:
    env = dict(os.environ, SCOUTER_IGEMM_TILE=t)
    print('=== tile', t or 'auto', flush=True)
    subprocess.run([sys.executable, '-c', code], env=env)
