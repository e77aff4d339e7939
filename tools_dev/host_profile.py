"""cProfile of the host side of one training step (tiny images: the GPU is never the limiter).  usage: python tools_dev/host_profile.py"""
import sys, time, torch, cProfile, pstats
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_model_gpu as T
from scouter_amd.optim import FusedAdamW
m, P, images, labels, cfg = T._synthetic_model("resnest26d", 10, 1, 3, 4, 64, 77)
opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
x, y = images.cuda(), labels.cuda()
def step():
    opt.zero_grad(); out, losses = m(x, y); losses[0].backward(m.loss_seed(losses[0])); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
