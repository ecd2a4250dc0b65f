#!/usr/bin/env python
"""probe (no profiler attached): GPU time between the end of the forward plan and the start of the backward plan of a steady-state
step -- the loss section holds ~190 us of kernels; what the GPU spends there beyond that is host starvation"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dualvar_amd import model as M
from dualvar_amd.optim import SGD
from dualvar_amd import engine as E
gpu = torch.device('cuda:0')
m = M.SimCLR_Naked('s3dg', 128, 0.07, False)
m.set_compute_dtype('fp32').train().to(gpu)
opt = SGD([p for p in m.parameters() if p.requires_grad], lr=0.003, momentum=0.9, weight_decay=1e-4, stores=m.stores())
block = torch.randn(64, 2, 3, 8, 112, 112, device=gpu)
ev = []
orig_f, orig_b = E.Plan.run_forward, E.Plan.run_backward
def rf(self):
    orig_f(self)
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(('f_end', e))
def rb(self):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(('b_start', e))
    orig_b(self)
E.Plan.run_forward, E.Plan.run_backward = rf, rb
def step():
    ret = m(block)
    loss = ret['clip_contrast_loss']
    opt.zero_grad()
    loss.backward()
    opt.step()
for _ in range(5):
    step()
torch.cuda.synchronize()
ev.clear()
t0 = time.perf_counter()
s0 = torch.cuda.Event(enable_timing=True); s0.record()
N = 20
for _ in range(N):
    step()
s1 = torch.cuda.Event(enable_timing=True); s1.record()
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = s0.elapsed_time(s1) / N
gaps = [a[1].elapsed_time(b[1]) for a, b in zip(ev[0::2], ev[1::2]) if a[0] == 'f_end' and b[0] == 'b_start']
print('step %.3f ms (host issue %.3f ms per step); loss section %.1f us mean, %.1f min, %.1f max over %d steps' % (
    tot, host * 1e3 / N, 1e3 * sum(gaps) / len(gaps), 1e3 * min(gaps), 1e3 * max(gaps), len(gaps)))
