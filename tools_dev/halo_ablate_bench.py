import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B=70
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out=[]
for cin, cout, g, H in [(128,256,2,56),(256,512,2,28),(512,1024,2,14)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(3, 3, cin // g, cout, device='cuda') * 0.05
    xp3 = K.planes_split(x, 3); wf3, wd3 = K.planes_split_weight(w, g, 3)
    out.append("%.0f/%.0f" % (timeit(lambda: K.conv2d_fwd_planes(xp3, wf3, 3, 3, 1, 1, g, tile=5)), timeit(lambda: K.conv2d_fwd_planes(xp3, wf3, 3, 3, 1, 1, g, tile=4))))
print("halo/tile4 us:", out)
