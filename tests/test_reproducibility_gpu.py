"""Every kernel family of the training step is launched repeatedly on the same inputs while the shared workspace and
the recycled output buffers are refilled with zeros / NaNs / large values in between: results must be bit-identical
(nothing read before it is written, no hand-off race between waves or launches, deterministic reductions).  The same
check caught a missing barrier in the fused xSlot backward (tests/test_xslot_gpu.py has that kernel's own variant)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def _poison(fill, dev):
    from scouter_amd import kernels as K
    K.workspace(1, dev).view(torch.float32).fill_(fill)
    junk = torch.empty(48 << 20, device=dev).fill_(fill)        # what torch.empty hands out next
    del junk


def _repeat(fn, dev, n=6):
    fills = [0.0, None, float("nan"), None, 3.0e30, None]
    outs = []
    for i in range(n):
        if fills[i % len(fills)] is not None:
            _poison(fills[i % len(fills)], dev)
        o = fn()
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (o if isinstance(o, (tuple, list)) else [o]) if torch.is_tensor(t)])
    for i, o in enumerate(outs[1:], 1):
        for a, b in zip(outs[0], o):
            assert torch.equal(torch.nan_to_num(a.float(), nan=7.0), torch.nan_to_num(b.float(), nan=7.0)), i
    for a in outs[0]:
        assert torch.isfinite(a.float()).all()


CONVS = [  # cin, cout, k, groups, H, stride, pad, B
    (64, 128, 1, 1, 28, 1, 0, 6), (64, 128, 3, 2, 28, 1, 1, 6), (32, 64, 3, 1, 40, 1, 1, 4), (128, 128, 3, 2, 14, 2, 1, 8),
    (256, 512, 1, 1, 7, 1, 0, 16),
]


@pytest.mark.parametrize("cfg", CONVS)
def test_convolution_kernels_are_reproducible(cfg):
    from scouter_amd import kernels as K
    cin, cout, k, g, H, s, p, B = cfg
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(cin + cout)
    x = torch.randn(B, H, H, cin, device=dev, generator=gen)
    w = torch.randn(k, k, cin // g, cout, device=dev, generator=gen) * 0.05
    y = K.conv2d_fwd(x, w, None, None, s, p, g)
    dy = torch.randn(y.shape, device=dev, generator=gen)
    add = torch.randn(x.shape, device=dev, generator=gen)
    _repeat(lambda: K.conv2d_fwd(x, w, None, None, s, p, g, False, True)[0], dev)
    _repeat(lambda: K.conv2d_fwd(x, w, None, None, s, p, g, False, True)[1][0], dev)        # the fused BN partial sums
    _repeat(lambda: K.conv2d_dgrad(dy, w, tuple(x.shape), add, s, p, g), dev)

    def wgrad():
        dw = torch.empty_like(w)
        K.conv2d_wgrad(x, dy, dw, s, p, g)
        return dw
    _repeat(wgrad, dev)


@pytest.mark.parametrize("shape", [(6, 28, 28, 128), (16, 7, 7, 512), (3, 1, 1, 64)])
def test_batchnorm_pool_and_split_attention_kernels_are_reproducible(shape):
    from scouter_amd import kernels as K
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(shape[1])
    C = shape[-1]
    x = torch.randn(shape, device=dev, generator=gen)
    res = torch.randn(shape, device=dev, generator=gen)
    gamma, beta = torch.rand(C, device=dev, generator=gen) + 0.5, torch.randn(C, device=dev, generator=gen) * 0.1

    def fwd():
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y, saved, mask = K.bn_fwd(x, gamma, beta, rm, rv, True, True, res, want_mask=True)
        return y, saved, mask, rm, rv
    _repeat(fwd, dev)
    y, saved, mask, _, _ = fwd()
    dy = torch.randn(shape, device=dev, generator=gen)

    def bwd():
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        dx, gout = K.bn_bwd(dy, None, x, saved, True, dg, db, True, mask=mask)
        return dx, gout, dg, db
    _repeat(bwd, dev)
    if shape[1] >= 7:
        _repeat(lambda: K.maxpool_fwd(x)[0], dev)
        yp, arg = K.maxpool_fwd(x)
        _repeat(lambda: K.maxpool_bwd(torch.ones_like(yp), arg, tuple(x.shape)), dev)
        _repeat(lambda: K.avgpool_fwd(x, 3, 2, 1, False, True), dev)
        ya = K.avgpool_fwd(x, 3, 2, 1, False, True)
        _repeat(lambda: K.avgpool_bwd(torch.ones_like(ya), tuple(x.shape), 3, 2, 1, False, True), dev)
    gap = K.sa_gap(x)
    _repeat(lambda: K.sa_gap(x), dev)
    a = K.radix_softmax_fwd(torch.randn(shape[0], C, device=dev, generator=gen))
    _repeat(lambda: K.sa_apply_fwd(x, a), dev)
    out = K.sa_apply_fwd(x, a)
    _repeat(lambda: K.sa_dattn(x, torch.ones_like(out)), dev)
    _repeat(lambda: K.sa_apply_bwd(torch.ones_like(out), a, torch.ones_like(gap)), dev)


def test_xslot_forward_is_reproducible():
    from scouter_amd import kernels as K
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(3)
    r = lambda *sh: torch.randn(*sh, device=dev, generator=gen)
    for B, S, N, T, L in ((3, 300, 49, 3, 3), (2, 96, 81, 3, 1), (4, 10, 49, 3, 3)):
        d = 64
        X, PE = r(B, N, d).relu_(), r(N, d) * 0.3
        tw, tb = [r(d, d) * 0.1 for _ in range(L)], [r(d) * 0.1 for _ in range(L)]
        s0 = r(S, d).abs() * 0.5
        gru = (r(3 * d, d) * 0.1, r(3 * d, d) * 0.1, r(3 * d) * 0.1, r(3 * d) * 0.1)

        def fwd():
            o = K.xslot_fwd(X, PE, tw, tb, s0, *gru, 1, T, 1)
            return o["logits"], o["attn"], o["area_part"], o["K"], o["H"], o["states"]
        _repeat(fwd, dev)


@pytest.mark.parametrize("arch,chan,img", [("resnest26d", 2048, 96), ("resnet18", 512, 64)])
def test_whole_training_steps_are_reproducible(arch, chan, img):
    """Four optimizer steps (forward, loss, backward with the weight gradients on the side stream, fused AdamW) from the
    same seed, twice, with differently poisoned allocator / workspace contents: every parameter, BatchNorm buffer and
    loss value must be bit-identical -- the step has no cross-stream race and no dependence on stale memory."""
    from scouter_amd.optim import FusedAdamW
    from scouter_amd.sloter.slot_model import SlotModel
    from scouter_amd.train import get_args_parser
    dev = torch.device("cuda")

    def run(fill):
        args = get_args_parser().parse_args(["--dataset", "ImageNet", "--model", arch, "--channel", str(chan), "--img_size",
                                             str(img), "--num_classes", "6", "--slots_per_class", "2", "--pre_trained",
                                             "false", "--lambda_value", "1"])
        for name, typ in (("num_classes", int), ("lambda_value", float), ("power", int), ("slots_per_class", int)):
            setattr(args, name, typ(getattr(args, name)))
        torch.manual_seed(11)
        model = SlotModel(args).cuda().train()
        opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
        g = torch.Generator().manual_seed(5)
        losses = []
        for it in range(4):
            x = torch.randn(6, 3, img, img, generator=g).cuda()
            y = torch.randint(0, 6, (6,), generator=g).cuda()
            _poison(fill if it % 2 == 0 else -fill, dev)
            opt.zero_grad()
            out, ls = model(x, y)
            ls[0].backward()
            opt.step()
            losses.append(ls[0].detach().clone())
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in model.state_dict().items()}, torch.stack(losses)

    sd0, l0 = run(0.0)
    sd1, l1 = run(float("nan"))
    sd2, l2 = run(1.0e30)
    assert torch.isfinite(l0).all()
    assert torch.equal(l0, l1) and torch.equal(l0, l2), (l0, l1, l2)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]) and torch.equal(sd0[k], sd2[k]), k
