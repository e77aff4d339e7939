"""Host time per training step (how far ahead of the GPU the launching thread runs): wall time of issuing 10 steps
without synchronising vs the GPU time of those steps."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from scouter_amd.optim import FusedAdamW
from scouter_amd.sloter.slot_model import SlotModel
cfg = dict(bench.CFG)
m = SlotModel(bench.make_args(cfg)).cuda().train()
opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
x = torch.randn(cfg["batch"], 3, 224, 224, device='cuda'); y = torch.randint(0, 10, (cfg["batch"],), device='cuda')
def step():
    opt.zero_grad(); out, losses = m(x, y); losses[0].backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host issue time %.2f ms/step, GPU-complete %.2f ms/step" % (t_issue * 100, t_all * 100))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
