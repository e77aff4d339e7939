"""Does the autograd-engine hop between forward and backward (root gradient fill, Function.backward, arena checks) leave the
GPU idle?  Times the bench step as written (loss.backward()) against the same kernels driven directly
(_forward_impl / _backward_impl, no autograd engine)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from scouter_amd.optim import FusedAdamW
from scouter_amd.sloter.slot_model import SlotModel
cfg = dict(bench.CFG)
torch.manual_seed(0)
m = SlotModel(bench.make_args(cfg)).cuda().train()
opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
x, y = bench.synth_batch(cfg["batch"], cfg["img_size"], cfg["num_classes"], 0, torch.device("cuda", 0))
one = torch.ones((), device="cuda")
def step_autograd():
    opt.zero_grad(); out, losses = m(x, y); losses[0].backward(); opt.step()
def step_direct():
    opt.zero_grad()
    m.grad_arena()
    logp, stats, state = m._forward_impl(x, y, save=True)
    m._backward_impl(state, None, one, None, None)
    opt.step()
def timeit(fn, n=60):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3):
    print("autograd %.3f ms   direct %.3f ms" % (timeit(step_autograd), timeit(step_direct)), flush=True)
