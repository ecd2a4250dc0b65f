#!/usr/bin/env python
"""probe: the backward launch list of the headline plan around the few-row data gradients, and the host time per launch"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dualvar_amd import model as M
gpu = torch.device('cuda:0')
m = M.SimCLR_Naked('s3dg', 128, 0.07, False)
m.set_compute_dtype('fp32').train().to(gpu)
block = torch.randn(64, 2, 3, 8, 112, 112, device=gpu)
for _ in range(2):
    ret = m(block)
    for st in m.stores():
        st.zero_grad()
    ret['clip_contrast_loss'].backward()
torch.cuda.synchronize()
pl = [p for lst in m.encoder_q[0]._plans.values() for p in lst][0]
names = [(type(l).__name__, l.name, getattr(l, 'kname', '')) for l in pl.b_list]
ks = [i for i, n in enumerate(names) if 'conv_gemm_ks<f32,DGRAD,64,32' in n[2]]
print(len(names), ks)
for i in range(0, min(60, len(names))):
    print(i, names[i])
# host cost of issuing the lists (GPU idle in between: pure host time)
for lst, nm in ((pl.f_list, 'forward'), (pl.b_list, 'backward')):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = torch.cuda.current_stream().cuda_stream
    for l in lst:
        l(s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(nm, len(lst), 'launches, host %.3f ms = %.1f us per launch' % ((t1 - t0) * 1e3, (t1 - t0) * 1e6 / len(lst)))
