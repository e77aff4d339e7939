import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_engine_gpu as T
from oracle import torch_oracle as O
from scouter_amd import engine
from scouter_amd.optim import FusedAdamW
from scouter_amd.sloter.slot_model import SlotModel
from scouter_amd.tools.calculate_tool import MetricLog
spec = O.state_dict_spec("resnet18", 10, 1, 1, in_chans=1, mnist_stem=True)
P0 = O.synth_state(spec, 300)
m = SlotModel(T._mnist_args()); m.load_state_dict(P0); m = m.cuda()
batches = [O.synth_batch(4, 1, 64, 10, 310 + i) for i in range(2)]
loader = [{"image": a.double(), "label": b} for a, b in batches]
opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
log = MetricLog()
engine.train_one_epoch(m, loader, opt, torch.device('cuda'), log.record, 0)
engine.evaluate(m, loader, torch.device('cuda'), log.record, 0)
print('hip record', log.record)
cfg = dict(model="resnet18", num_classes=10, slots_per_class=1, loss_status=1, power=1, lambda_value=1.0)
# oracle trained (fp64)
P = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in O.synth_state(spec, 300).items()}
tr = O.OracleTrainer(P, cfg, lr=1e-4)
for a, b in batches: tr.step(a.double(), b)
mine = {k: v.detach().cpu() for k, v in m.state_dict().items()}
worst = []
for k in mine:
    if not mine[k].dtype.is_floating_point: continue
    d = (mine[k].double() - tr.P[k]).abs()
    worst.append((float(d.max()), float((d > 5e-5).double().mean()), k))
for w in sorted(worst, reverse=True)[:12]: print('param diff max %.3e frac>5e-5 %.4f %s' % w)
# oracle eval on MY params vs on oracle params
def ev(Pd):
    tot = 0
    with torch.no_grad():
        for a, b in batches:
            out, losses = O.slot_model_forward(Pd, a.double(), b, cfg, training=False)
            tot += float(losses[0])
    return tot / 2
mineP = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in mine.items()}
print('oracle eval on HIP params', ev(mineP), ' oracle eval on oracle params', ev(tr.P))
k = 'backbone.layer3.0.conv1.weight'
d = (mine[k].double() - tr.P[k]).abs()
phys = d.permute(2, 3, 1, 0).reshape(-1)       # HWIO physical order
idx = torch.nonzero(phys > 5e-5).flatten()
print(k, 'numel', phys.numel(), 'bad', idx.numel(), 'first idx', idx[:20].tolist(), 'last', idx[-5:].tolist())
print('bad by cout (top):', torch.bincount(idx % d.shape[0], minlength=d.shape[0]).topk(8))
print('bad by cin (top):', torch.bincount((idx // d.shape[0]) % d.shape[1], minlength=d.shape[1]).topk(8))
# after ONE step only
m2 = SlotModel(T._mnist_args()); m2.load_state_dict(P0); m2 = m2.cuda().train()
opt2 = FusedAdamW([p for p in m2.parameters() if p.requires_grad], lr=1e-4)
a, b = batches[0]
out, losses = m2(a.cuda(), b.cuda()); losses[0].backward(); opt2.step(); torch.cuda.synchronize()
P1 = {kk: (v.double() if v.dtype.is_floating_point else v.clone()) for kk, v in O.synth_state(spec, 300).items()}
tr1 = O.OracleTrainer(P1, cfg, lr=1e-4); tr1.step(a.double(), b)
d1 = (m2.state_dict()[k].cpu().double() - tr1.P[k]).abs()
print('after ONE step: max %.3e frac>5e-5 %.5f' % (float(d1.max()), float((d1 > 5e-5).double().mean())))
g_hip = dict(m2.named_parameters())[k].grad.detach().cpu().double()
P1b = {kk: (v.double() if v.dtype.is_floating_point else v.clone()) for kk, v in O.synth_state(spec, 300).items()}
trb = O.OracleTrainer(P1b, cfg, lr=1e-4)
_, _, _, gref = trb.step(a.double(), b)
g64 = gref[k]
bad = torch.nonzero((d1 > 5e-5)).tolist()[:10]
for ix in bad:
    ix = tuple(ix)
    print(ix, 'g_hip % .3e  g_fp64 % .3e   p_hip %.7f p_ref %.7f p0 %.7f' % (float(g_hip[ix]), float(g64[ix]), float(m2.state_dict()[k].cpu()[ix]), float(tr1.P[k][ix]), float(P0[k][ix])))
print('grad scale', float(g64.abs().max()), 'median |g|', float(g64.abs().median()))
rel = ((g_hip - g64).abs() / g64.abs().clamp_min(1e-30))
print('frac rel err > 0.5:', float((rel > 0.5).double().mean()), ' frac |g64| < 1e-7:', float((g64.abs() < 1e-7).double().mean()))
