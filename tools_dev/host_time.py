"""Pure host cost of one training step: tiny images so that the GPU is never the limiter."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from scouter_amd.optim import FusedAdamW
from scouter_amd.sloter.slot_model import SlotModel
cfg = dict(bench.CFG)
m = SlotModel(bench.make_args(cfg)).cuda().train()
opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
x = torch.randn(2, 3, 64, 64, device='cuda'); y = torch.randint(0, 10, (2,), device='cuda')
def step():
    opt.zero_grad(); out, losses = m(x, y); losses[0].backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("host-bound step: %.2f ms" % ((time.perf_counter() - t0) * 50))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
