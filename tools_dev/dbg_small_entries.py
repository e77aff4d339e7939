import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_model_gpu as T
case = 'resnet18_mnist_64'
m, P, images, labels = T.build(case)
m.train()
out, (loss, nll, area) = m(images.cuda(), labels.cuda()); loss.backward(); torch.cuda.synchronize()
_, _, _, l64, _ = T.oracle_run(case, torch.float64)
_, _, _, l32, _ = T.oracle_run(case, torch.float32)
named = dict(m.named_parameters())
for k in ['backbone.layer3.0.conv1.weight', 'backbone.layer4.0.conv2.weight', 'backbone.layer1.1.conv1.weight', 'backbone.layer4.0.downsample.0.weight', 'backbone.layer1.0.conv1.weight', 'conv1x1.weight', 'backbone.conv1.weight']:
    r = l64[k].grad; a = named[k].grad.detach().cpu().double(); b = l32[k].grad.double()
    sc = float(r.abs().max())
    ea, eb = (a - r).abs(), (b - r).abs()
    print('%-40s scale %.2e | hip: max %.2e  mean %.2e  bad(>50%%rel) %.5f | torch32: max %.2e mean %.2e bad %.5f' % (
        k, sc, float(ea.max()) / sc, float(ea.mean()) / sc, float((ea > 0.5 * r.abs()).double().mean()),
        float(eb.max()) / sc, float(eb.mean()) / sc, float((eb > 0.5 * r.abs()).double().mean())))
    # error by tap
    if r.dim() == 4 and r.shape[-1] == 3:
        print('   hip mean err by tap:', (ea.mean(dim=(0, 1)) / sc).numpy().round(9).tolist())
