"""Throughput / memory smoke of the other BASELINE configs at their full per-GPU batch (1 GPU).
usage: python tools_dev/config_sweep.py [c1 c3 c4 c5 c2_260 c2 ...] [--bf16]"""
import argparse, sys, time, torch
sys.path.insert(0, '.')
import bench
from scouter_amd.optim import FusedAdamW
from scouter_amd.sloter.slot_model import SlotModel
CONFIGS = {
    'c1': dict(model='resnet18', num_classes=10, slots_per_class=1, channel=512, to_k_layer=1, power=1, img=224, batch=64, mnist=True),
    'c3': dict(model='resnest26d', num_classes=10, slots_per_class=1, channel=2048, to_k_layer=3, power=2, img=224, batch=70, loss_status=-1),
    'c4': dict(model='resnest26d', num_classes=200, slots_per_class=1, channel=2048, to_k_layer=3, power=2, img=224, batch=128),
    'c5': dict(model='resnest50d', num_classes=100, slots_per_class=3, channel=2048, to_k_layer=3, power=2, img=224, batch=256),
    'c2_260': dict(model='resnest26d', num_classes=10, slots_per_class=1, channel=2048, to_k_layer=3, power=2, img=260, batch=70),
}
CONFIGS['c2'] = dict(model='resnest26d', num_classes=10, slots_per_class=1, channel=2048, to_k_layer=3, power=2, img=224, batch=70)
BF16 = '--bf16' in sys.argv
names = [a for a in sys.argv[1:] if not a.startswith('--')]
for name in names or list(CONFIGS):
    c = CONFIGS[name]
    cfg = dict(bench.CFG, **{k: v for k, v in c.items() if k in bench.CFG})
    a = bench.make_args(cfg)
    a.precision = 'bf16' if BF16 else 'fp32'
    if c.get('mnist'): a.dataset = 'MNIST'
    torch.manual_seed(0)
    m = SlotModel(a).cuda().train()
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    x = torch.randn(c['batch'], 1 if c.get('mnist') else 3, c['img'], c['img'], device='cuda')
    y = torch.randint(0, c['num_classes'], (c['batch'],), device='cuda')
    def step():
        opt.zero_grad(); out, losses = m(x, y); losses[0].backward(); opt.step(); return losses
    for _ in range(3): losses = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): losses = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(('bf16 ' if BF16 else 'fp32 ') + '%-7s %-11s C=%-3d spc=%d B=%-3d %dx%d : %7.1f img/s  %6.2f ms/step  loss %.4f  peak mem %.1f GB' % (
        name, c['model'], c['num_classes'], c['slots_per_class'], c['batch'], c['img'], c['img'], c['batch'] / dt, dt * 1e3,
        float(losses[0]), torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del m, opt, x, y; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
