#!/usr/bin/env python
"""probe: bandwidth-bound vs matrix-bound time estimate of the backward main chain of a plan"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dualvar_amd import model as M
from dualvar_amd import engine as E
gpu = torch.device('cuda:0')
for net, B, T, S in (('s3dg', 64, 8, 112), ('r3d', 32, 8, 112), ('r21d', 32, 8, 112), ('r50', 4, 32, 224), ('s3dg', 64, 16, 112)):
    m = M.SimCLR_Naked(net, 128, 0.07, False)
    m.set_compute_dtype('fp32').train().to(gpu)
    block = torch.randn(B, 2, 3, T, S, S, device=gpu)
    ret = m(block)
    pl = [p for lst in m.encoder_q[0]._plans.values() for p in lst][0]
    th = tm = tw = 0.0
    for l in pl.b_list:
        if l.name in E.SIDE_LAUNCHES:
            tw += max(l.flops / 1.5e14, l.bytes / 4e12)
        elif l.name == 'conv_dgrad':
            tm += max(l.flops / 1.5e14, l.bytes / 4e12)
        else:
            th += l.bytes / 4e12
    print('%-5s T%-2d: main chain conv %.2f ms, other %.2f ms, ratio %.2f; side %.2f ms' % (net, T, tm * 1e3, th * 1e3, th / max(tm, 1e-9), tw * 1e3))
    del m, ret, pl
    torch.cuda.empty_cache()
