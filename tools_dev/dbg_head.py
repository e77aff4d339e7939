import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import torch_oracle as O
from oracle.gen_golden import HEAD_CASES, head_inputs
from scouter_amd import kernels as K
case = sys.argv[1] if len(sys.argv) > 1 else 'c1_mnist'
C, spc, side, L, ls, power, B, Cin = HEAD_CASES[case]
feat, labels, P = head_inputs(case)
Pd = {k: v.double() for k, v in P.items()}
aux = {}
out, losses = O.head_forward(Pd, feat.double(), labels, dict(num_classes=C, slots_per_class=spc, loss_status=ls, power=power, lambda_value=1.0), aux=aux)
X = aux['x'].float().contiguous().cuda()     # [B,N,64] oracle tokens
PE = K.posenc_sine(side, side, 64, X.device)
tw = [P['slot.to_k.%d.weight' % (2*l)].cuda() for l in range(L)]
tb = [P['slot.to_k.%d.bias' % (2*l)].cuda() for l in range(L)]
o = K.xslot_fwd(X, PE, tw, tb, P['slot.initial_slots'][0].contiguous().cuda(), P['slot.gru.weight_ih_l0'].cuda(), P['slot.gru.weight_hh_l0'].cuda(), P['slot.gru.bias_ih_l0'].cuda(), P['slot.gru.bias_hh_l0'].cuda(), spc, 3, ls)
torch.cuda.synchronize()
def e(a, b, n): print('%-10s max|diff| %.3e   ref max %.3e' % (n, float((a.cpu().double() - b).abs().max()), float(b.abs().max())))
e(o['K'], aux['k'], 'K')
pe_ref = O.posenc_sine(side, side, 64).reshape(64, -1).t().double()
e(o['H'][0], aux['x'] + pe_ref, 'H0')
for t in range(2): e(o['states'][t], aux['slot_states'][t+1], 'state%d' % (t+1))
e(o['attn'], aux['attn'], 'attn')
e(o['logits'], aux['logits'], 'logits')
print('area', float(o['area_part'].sum()), float(aux['attn'].sum()))
