#!/usr/bin/env python
"""probe: the same forward + backward THREE times through one plan (no optimizer step): every pass must give the first pass's
gradients bit for bit -- the shared ticket workspace of the fused BatchNorm-backward reduce is reused across passes"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dualvar_amd import model as M
net, clips = (sys.argv[1] if len(sys.argv) > 1 else 's3dg'), int(sys.argv[2]) if len(sys.argv) > 2 else 16
gpu = torch.device('cuda:0')
block = torch.randn(clips, 2, 3, 8, 112, 112, generator=torch.Generator().manual_seed(3)).to(gpu)
torch.manual_seed(0)
m = M.SimCLR_Naked(net, 128, 0.07, False)
m.set_compute_dtype('fp32').train().to(gpu)
grads = []
for it in range(3):
    ret = m(block)
    for st in m.stores():
        st.zero_grad()
    ret['clip_contrast_loss'].backward()
    torch.cuda.synchronize()
    grads.append(torch.cat([st.grad.detach().float().flatten().clone() for st in m.stores()]))
    print('pass', it, 'loss', float(ret['clip_contrast_loss'].detach()), '|g|', float(grads[-1].norm()), flush=True)
for it in (1, 2):
    d = float((grads[it] - grads[0]).abs().max())
    print('pass %d vs pass 0: max |diff| %.3e of %.3e' % (it, d, float(grads[0].abs().max())))
plans = [pl for lst in m.encoder_q[0]._plans.values() for pl in lst]
print('fused reduces:', sum(1 for pl in plans for op in pl.ops if getattr(op, 'bn_fuse_tap', False)))
