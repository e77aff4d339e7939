"""Phase times of the fp32 1x1 weight-gradient kernel per workgroup (-DIGEMM_STAMPS build; 100 MHz wall clock): operands
(until the first fragments), chunk loop, epilogue (slab store retired), chunks per workgroup, resident workgroups per CU.
usage: SCOUTER_HIP_LIB=build_dev/libscouter_igs.so python tools_dev/wgrad_stamps.py"""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K, _native
L = _native.lib()
L.scouter_dev_set_igemm_stamps.argtypes = [ctypes.c_void_p]
B = 70
for cin, cout, H in [(256, 64, 56), (64, 256, 56), (128, 512, 28), (512, 128, 28), (256, 1024, 14), (1024, 256, 14), (512, 2048, 7),
                     (1024, 2048, 7), (2048, 512, 7)]:
    x = torch.randn(B, H, H, cin, device='cuda'); dy = torch.randn(B, H, H, cout, device='cuda')
    dw = torch.empty(1, 1, cin, cout, device='cuda')
    st = torch.zeros(8 * 70000, dtype=torch.int64, device='cuda')
    for _ in range(3):
        K.conv2d_wgrad(x, dy, dw, 1, 0, 1)
    torch.cuda.synchronize()
    L.scouter_dev_set_igemm_stamps(ctypes.c_void_p(st.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.conv2d_wgrad(x, dy, dw, 1, 0, 1); e1.record(); torch.cuda.synchronize()
    L.scouter_dev_set_igemm_stamps(None)
    s = st.view(-1, 8).cpu()
    s = s[s[:, 3] > 0].double()
    t0 = s[:, 0].min()
    pro, loop, epi, tot = (s[:, 1] - s[:, 0]) / 100, (s[:, 2] - s[:, 1]) / 100, (s[:, 3] - s[:, 2]) / 100, (s[:, 3] - s[:, 0]) / 100
    span = (s[:, 3].max() - t0) / 100
    kt = s[:, 5].mean()
    fl = 2.0 * B * H * H * cin * cout
    plan = K._tile_cache.get(("wgrad", False, B, H, H, cin, cout, 1, 1, 1, 0, 1))
    print("wgrad %4d->%4d @%2d plan %s: %5d workgroups x %.0f chunks, kernel+reduce %.1f us (MFMA floor %.1f); per workgroup: operands "
          "%.2f us, chunk loop %.2f us (%.0f ns per chunk; 16 MFMAs = 427 ns at one wave per SIMD), epilogue %.2f us; mean resident "
          "workgroups per CU %.2f" % (cin, cout, H, plan, len(s), kt, e0.elapsed_time(e1) * 1e3, fl / 157.3e6, pro.mean(), loop.mean(),
                                      1e3 * loop.mean() / kt, epi.mean(), float(tot.sum()) / (256 * span)))
