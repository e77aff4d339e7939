"""Phase times inside phalo_kernel from a -DPHALO_ABLATE=8 build (tools_dev/phalo_ablate.sh 8): per workgroup the cycle
counter at start / prologue done / K loop done / stores retired.  usage: SCOUTER_HIP_LIB=build_dev/libscouter_ab8.so
python tools_dev/halo_stamps.py"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K, _native
B = 70
for cin, cout, g, H in [(128, 256, 2, 56), (256, 512, 2, 28), (512, 1024, 2, 14)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(3, 3, cin // g, cout, device='cuda') * 0.05
    xp3 = K.planes_split(x, 3); wf3, _ = K.planes_split_weight(w, g, 3)
    y = torch.empty(B, H, H, cout, device='cuda')
    nblk = -(-B * H * H // 256) * (cout // g // 128) * g
    st = torch.zeros(nblk * 4 + 64, dtype=torch.int64, device='cuda')
    for _ in range(3):
        _native.check(_native.lib().scouter_conv2d_fwd_planes(xp3.data_ptr(), wf3.data_ptr(), None, None, y.data_ptr(), st.data_ptr(),
                      B, H, H, cin, cout, 3, 3, 1, 1, g, 0, 3, 5, None), "x")
    torch.cuda.synchronize()
    s = st[:nblk * 4].view(nblk, 4).cpu().double()
    t0 = s[:, 0].min()
    pro, loop, epi = (s[:, 1] - s[:, 0]), (s[:, 2] - s[:, 1]), (s[:, 3] - s[:, 2])
    span = (s[:, 3].max() - t0)
    steps = 9 * (cin // g) // 16
    print("%s: %d workgroups; cycles (100 MHz counter?) prologue %.0f  K loop %.0f (%.1f per step)  epilogue+stores %.0f ; whole kernel %.0f ; sum/CU-rounds %.2f"
          % ((cin, cout, g, H), nblk, pro.mean(), loop.mean(), loop.mean() / steps, epi.mean(), span,
             float((s[:, 3] - s[:, 0]).sum() / 256 / span)))
