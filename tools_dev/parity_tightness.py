"""How tight is the whole-model parity really?  Per fixture case: |HIP - fp64| next to |torch fp32 - fp64| on the same
inputs (log-probs, attention), and per parameter tensor the gradient error next to plain fp32 PyTorch's (oracle fp32 vs
oracle fp64) -- the numbers behind the assertions of tests/test_model_gpu.py."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_model_gpu import build, oracle_run, capture_relu_signs, GOLD
for case in sys.argv[1:] or ["resnet18_mnist_64", "resnest26d_96", "resnest50d_64_spc3", "resnest26d_224"]:
    g = np.load(os.path.join(GOLD, "model_%s.npz" % case))
    m, P, images, labels = build(case)
    m.train()
    signs = capture_relu_signs(m)
    out, (loss, nll, area) = m(images.cuda(), labels.cuda())
    loss.backward(); torch.cuda.synchronize()
    floor = float(np.abs(g["f32_log_probs"] - g["f64_log_probs"]).max())
    err = float(np.abs(out.detach().cpu().numpy() - g["f64_log_probs"]).max())
    ea = float(np.abs(m.slot.last_attn.cpu().numpy() - g["f64_attn"]).max())
    fa = float(np.abs(g["f32_attn"] - g["f64_attn"]).max()) if "f32_attn" in g else float('nan')
    print("%-22s log_probs: HIP %.3g torch32 %.3g ratio %.2f | attn: HIP %.3g torch32 %.3g" % (case, err, floor, err / max(floor, 1e-12), ea, fa))
    if case == "resnest26d_224":
        continue
    pool_arg = signs.pop("maxpool").permute(0, 3, 1, 2).cpu()
    masks = {k: (v > 0).permute(0, 3, 1, 2).cpu() for k, v in signs.items()}
    own = {}
    _, _, _, leaves, _ = oracle_run(case, torch.float64, relu_masks=masks, sign_log=own, pool_arg=pool_arg)
    _, _, _, leaves32, _ = oracle_run(case, torch.float32, relu_masks=masks, pool_arg=pool_arg)
    print("   ReLU flips HIP vs oracle fp64:", {k: int((own[k] != masks[k]).sum()) for k in masks if int((own[k] != masks[k]).sum())})
    named = dict(m.named_parameters())
    viol = []
    for k, ref in leaves.items():
        if k.endswith("conv2.fc1.bias"):
            continue
        mine = named[k].grad.detach().cpu().double()
        r = ref.grad
        scale = float(r.abs().max())
        e = float((mine - r).abs().max()); e32 = float((leaves32[k].grad.double() - r).abs().max())
        if e > max(2 * e32, 1e-3 * scale):
            viol.append((e / max(scale, 1e-30), k, e, e32, scale))
    viol.sort(reverse=True)
    print("   gradient tensors violating e <= max(2 e32, 1e-3 scale): %d of %d" % (len(viol), len(leaves)))
    for v in viol[:12]:
        print("     rel %.3g  %-50s e %.3g e32 %.3g scale %.3g" % v)
