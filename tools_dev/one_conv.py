import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
cin, cout, k, g, H, B = 128, 256, 3, 2, 56, 70
mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
y = K.conv2d_fwd(x, w, None, None, 1, 1, g); dy = torch.randn_like(y); dw = torch.empty_like(w)
for _ in range(5):
    if mode == 'fwd': K.conv2d_fwd(x, w, None, None, 1, 1, g)
    elif mode == 'dgrad': K.conv2d_dgrad(dy, w, tuple(x.shape), None, 1, 1, g)
    else: K.conv2d_wgrad(x, dy, dw, 1, 1, g)
torch.cuda.synchronize()
