import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_model_gpu as T
case = sys.argv[1]
import os
if os.environ.get('SEED'):
    _orig = T.model_inputs
    T.model_inputs = lambda c: _orig(c, seed=int(os.environ['SEED']))
m, P, images, labels = T.build(case)
m.train()
out, (loss, nll, area) = m(images.cuda(), labels.cuda())
loss.backward(); torch.cuda.synchronize()
_, _, _, leaves, Q = T.oracle_run(case, torch.float64)
_, _, _, l32, _ = T.oracle_run(case, torch.float32)
named = dict(m.named_parameters())
rows = []
for k, ref in leaves.items():
    r = ref.grad; mine = named[k].grad.detach().cpu().double()
    sc = float(r.abs().max()) + 1e-12
    rows.append((float((mine - r).abs().max()) / sc, float((l32[k].grad.double() - r).abs().max()) / sc, sc, k))
for e, e32, sc, k in rows[:40]: print('%-45s hip %.2e  torch32 %.2e  scale %.2e' % (k, e, e32, sc))
print('...')
for e, e32, sc, k in sorted(rows, reverse=True)[:12]: print('WORST %-45s hip %.2e  torch32 %.2e  scale %.2e' % (k, e, e32, sc))
if len(sys.argv) > 2:
    k = sys.argv[2]
    d = (named[k].grad.detach().cpu().double() - leaves[k].grad)
    print(k, 'err by tap (kh,kw):'); print(d.abs().amax(dim=(0, 1)))
    print('err by cout:', d.abs().amax(dim=(1, 2, 3))[:16])
    print('err by cin:', d.abs().amax(dim=(0, 2, 3))[:16])
print('---- reverse order (head first)')
for e, e32, sc, k in rows[::-1][:90]:
    if 'fc1.bias' in k: continue
    print('%-45s hip %.2e  torch32 %.2e' % (k, e, e32))
