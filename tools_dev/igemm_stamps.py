"""Phase times of the fp32 1x1 kernels per workgroup (-DIGEMM_STAMPS build; 100 MHz wall clock): start -> first fragments
(operand latency), K loop, epilogue (LDS staging, BatchNorm statistics, stores retired), and how many workgroups a CU holds
at a time.  usage: SCOUTER_HIP_LIB=build_dev/libscouter_igs.so python tools_dev/igemm_stamps.py"""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K, _native
L = _native.lib()
L.scouter_dev_set_igemm_stamps.argtypes = [ctypes.c_void_p]
B = 70
for cin, cout, H, mode, tile in [(64, 256, 56, "fwd", 0), (64, 256, 56, "fwd", 2), (256, 64, 56, "fwd", 2), (128, 512, 28, "fwd", 2),
                                 (512, 2048, 7, "fwd", 2), (1024, 2048, 7, "fwd", 2), (256, 64, 56, "dgrad", 2), (64, 256, 56, "dgrad", 2)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    dy = torch.randn(B, H, H, cout, device='cuda')
    key = (mode, False, B, H, H, cin, cout, 1, 1, 1, 0, 1)
    K._tile_cache[key] = tile
    st = torch.zeros(8 * 70000, dtype=torch.int64, device='cuda')

    def run():
        if mode == "fwd":
            return K.conv2d_fwd(x, w, None, None, 1, 0, 1, bn_stats=True)
        return K.conv2d_dgrad(dy, w, tuple(x.shape), None, 1, 0, 1)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.scouter_dev_set_igemm_stamps(ctypes.c_void_p(st.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    L.scouter_dev_set_igemm_stamps(None)
    s = st.view(-1, 8).cpu()
    s = s[s[:, 3] > 0].double()
    t0 = s[:, 0].min()
    pro, loop, epi, tot = (s[:, 1] - s[:, 0]) / 100, (s[:, 2] - s[:, 1]) / 100, (s[:, 3] - s[:, 2]) / 100, (s[:, 3] - s[:, 0]) / 100
    span = (s[:, 3].max() - t0) / 100
    # workgroups alive at the same time on one CU: sum of lifetimes / (CUs x span)
    cus = len(set(s[:, 4].long().tolist()))
    print("%-6s %4d->%4d @%2d tile %d: %5d workgroups, kernel %.1f us (event %.1f); per workgroup: operands %.2f us, K loop %.2f us, "
          "epilogue %.2f us, total %.2f us; %d distinct CU ids, mean resident workgroups per CU %.2f"
          % (mode, cin, cout, H, tile, len(s), span, e0.elapsed_time(e1) * 1e3, pro.mean(), loop.mean(), epi.mean(), tot.mean(),
             cus, float(tot.sum()) / (256 * span)))
