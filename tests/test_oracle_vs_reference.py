"""Live pin of the restatement against the reference itself.  Runs only where /root/reference exists (the build
container); on the GPU box it is skipped and the committed fixtures (test_oracle_golden.py) carry the pin."""
import numpy as np
import pytest
import torch

from oracle import ref_import as R
from oracle import torch_oracle as O

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="reference tree not present")


@pytest.mark.parametrize("arch,C,spc,L,mnist", [("resnet18", 10, 1, 1, True), ("resnest26d", 10, 1, 3, False),
                                               ("resnest50d", 100, 3, 3, False)])
def test_state_dict_spec_equals_reference(arch, C, spc, L, mnist):
    args = R.make_args(model=arch, num_classes=C, slots_per_class=spc, channel=O.ARCHS[arch]["channel"],
                       to_k_layer=L, dataset="MNIST" if mnist else "ImageNet")
    sd = R.build_reference_slot_model(args).state_dict()
    spec = O.state_dict_spec(arch, C, spc, L, in_chans=1 if mnist else 3, mnist_stem=mnist)
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in sd)


@pytest.mark.parametrize("C,spc,N,L,ls,power", [(10, 1, 49, 3, 1, 2), (6, 3, 81, 1, -1, 1)])
def test_xslot_restatement_equals_reference_module(C, spc, N, L, ls, power):
    m = R.build_reference_slot_attention(C, spc, 64, loss_status=ls, power=power, to_k_layer=L)
    P = {"slot." + k: v.detach() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(5)
    x = torch.from_numpy(np.maximum(rng.standard_normal((3, N, 64)), 0).astype(np.float32))
    pe = torch.from_numpy(rng.standard_normal((1, N, 64)).astype(np.float32))
    with torch.no_grad():
        ref_logits, ref_loss = m(x + pe, x)
        logits, loss = O.xslot_forward(P, x + pe, x, C, spc, ls, power)
    np.testing.assert_allclose(logits.numpy(), ref_logits.numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(float(loss), float(ref_loss), atol=1e-7, rtol=1e-6)
