#!/usr/bin/env python
"""probe: gradients of one R(2+1)D TimeSeriesV4 fixture step -> a file (run under different env switches, then compare)"""
import os, sys, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import procedural as P
from dualvar_amd import model as M
from tests.util import CLIP, total_loss
out = sys.argv[1]
net = sys.argv[2] if len(sys.argv) > 2 else 'r21d'
gpu = torch.device('cuda:0')
torch.manual_seed(0)
m = M.SimCLR_TimeSeriesV4(net, 128, 0.07, False, args=types.SimpleNamespace(shufflerank_theta=0.05))
P.procedural_init(m)
m.set_compute_dtype('fp32').train().to(gpu)
block = P.procedural_clips(2, 3, **CLIP).to(gpu)
np.random.seed(1234)
ret = m(block)
loss = total_loss(ret)
for st in m.stores():
    st.zero_grad()
loss.backward()
torch.cuda.synchronize()
d = {n: p.grad.detach().float().cpu().numpy() for n, p in m.named_parameters() if p.grad is not None}
d['__loss'] = np.array(float(loss.detach()))
np.savez(out, **d)
print('loss', float(loss.detach()), len(d))
