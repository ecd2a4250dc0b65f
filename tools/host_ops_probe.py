#!/usr/bin/env python
"""Which torch ops (copies, fills, small elementwise kernels) one training step still issues next to the library launches, by
source line (development aid).    python tools/host_ops_probe.py [--dtype fp32|bf16] [--model simclr_naked]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--net', default='s3dg')
    ap.add_argument('--loop', type=int, default=0, help='just run this many steps (for rocprofv3 --kernel-trace + tools/timeline_gaps.py)')
    args = ap.parse_args()
    from dualvar_amd import model as M
    from dualvar_amd.optim import SGD
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = M.SimCLR_Naked(args.net, 128, 0.07, False)
    m.set_compute_dtype(args.dtype).train().to(dev)
    opt = SGD([p for p in m.parameters() if p.requires_grad], lr=0.003, momentum=0.9, weight_decay=1e-4, stores=m.stores())
    x = torch.randn(64, 3, 16, 112, 112, device=dev)

    def step():
        block = x.view(64, 3, 2, 8, 112, 112).transpose(1, 2)
        ret = m(block)
        loss = sum(v for k, v in ret.items() if 'loss' in k)
        opt.zero_grad()
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if args.loop:
        for _ in range(args.loop):
            step()
        torch.cuda.synchronize()
        return
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                                with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    for ev in prof.events():
        if ev.name.startswith('aten::') and ev.name in ('aten::copy_', 'aten::zero_', 'aten::fill_', 'aten::clone', 'aten::contiguous',
                                                      'aten::add_', 'aten::mul_', 'aten::add', 'aten::mul', 'aten::sum', 'aten::cat',
                                                      'aten::index', 'aten::to', 'aten::_to_copy', 'aten::zeros', 'aten::randperm'):
            st = [s for s in (ev.stack or []) if 'dualvar_amd' in s or 'host_ops_probe' in s or 'bench.py' in s]
            cnt[(ev.name, st[0] if st else '?')] += 1
    for (name, where), n in cnt.most_common(60):
        print('%4d  %-18s %s' % (n, name, where))
    kc = collections.Counter(ev.name for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    print('--- device activities')
    for k, n in kc.most_common(25):
        print('%4d  %s' % (n, k[:110]))


if __name__ == '__main__':
    main()
