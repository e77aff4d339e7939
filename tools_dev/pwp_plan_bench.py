"""Plans of the exact-fp32 persistent pointwise kernel (tile 4): columns per workgroup and resident workgroups per CU, through the
development override SCOUTER_PWP_PLAN_DEV="bn,resident" (csrc/conv_igemm.hip pwp_plan).  One process per plan (the env is read at
every launch, so one process would do; separate lines keep it simple).  usage: python tools_dev/pwp_plan_bench.py [B]"""
import os, sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout, H in [(64, 256, 56), (64, 64, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    key = ("fwd", False, B, H, H, cin, cout, 1, 1, 1, 0, 1)
    out = []
    ref = None
    for plan in ("", "256,1", "128,1", "64,1", "64,2", "128,2", "64,3"):
        if plan: os.environ["SCOUTER_PWP_PLAN_DEV"] = plan
        else: os.environ.pop("SCOUTER_PWP_PLAN_DEV", None)
        K._tile_cache[key] = 4
        try:
            t = timeit(lambda: K.conv2d_fwd(x, w, bn_stats=True))
            y = K.conv2d_fwd(x, w)
            ref = y if ref is None else ref
            out.append("%s: %.1f%s" % (plan or "default", t, "" if torch.equal(y, ref) else " (BITS DIFFER)"))
        except RuntimeError as e:
            out.append("%s: n/a" % plan)
    K._tile_cache[key] = 2
    os.environ.pop("SCOUTER_PWP_PLAN_DEV", None)
    t2 = timeit(lambda: K.conv2d_fwd(x, w, bn_stats=True))
    print("%-16s tile 2 %.1f us | tile 4 by plan (bn,resident) us: %s" % (str((cin, cout, H)), t2, "  ".join(out)))
