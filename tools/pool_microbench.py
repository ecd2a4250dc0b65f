#!/usr/bin/env python
"""Time dv_maxpool3d_fwd / dv_maxpool3d_bwd on the pools of the S3D-G pretrain step (per-GPU batch 128 x 8 x 112 x 112),
and a known-byte copy next to them (calibration of the FETCH_SIZE / WRITE_SIZE PMC counters: `--calib`).

    python tools/pool_microbench.py [--dtype fp32|bf16] [--only p3c] [--passes fwd,bwd] [--reps 20] [--calib]

Prints one line per (pool, pass): microseconds and algorithmic GB/s (fwd: read x, write y + idx; bwd: read dy + idx,
write dx).  A development aid for the GPU box; not part of the product path or the tests."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualvar_amd import _lib as L          # noqa: E402
from dualvar_amd import ops                # noqa: E402

# (N, T, H, W, C, k, s, p)
POOLS = {
    'pool2a': (128, 4, 56, 56, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    'pool3a': (128, 4, 28, 28, 192, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    'p3b': (128, 4, 14, 14, 192, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    'p3c': (128, 4, 14, 14, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    'pool4a': (128, 4, 14, 14, 480, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    'p4b': (128, 2, 7, 7, 480, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    'p4e': (128, 2, 7, 7, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    'p4f': (128, 2, 7, 7, 528, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    'pool5a': (128, 2, 7, 7, 832, (2, 2, 2), (2, 2, 2), (0, 0, 0)),
    'p5b': (128, 1, 3, 3, 832, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # 16-frame clips (BASELINE cfg 2), batch 64
    'p3c_t16': (64, 8, 14, 14, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
}


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--passes', default='fwd,bwd')
    ap.add_argument('--dtype', default='fp32', choices=['bf16', 'fp32'])
    ap.add_argument('--calib', action='store_true', help='also run a 256 MiB device copy (known bytes for the PMC counters)')
    args = ap.parse_args()
    L.require_device()
    dev = torch.device('cuda:0')
    dt = L.DV_F32 if args.dtype == 'fp32' else L.DV_BF16
    es = 4 if args.dtype == 'fp32' else 2
    tdt = torch.float32 if args.dtype == 'fp32' else torch.bfloat16
    if args.calib:
        src = torch.randn(64 << 20, device=dev)
        dst = torch.empty_like(src)
        us = timed(lambda: dst.copy_(src), args.reps)
        print('calib copy 256 MiB read + 256 MiB write: %8.1f us  %7.1f GB/s' % (us, 2 * src.numel() * 4 / us / 1e3))
    names = [n for n in POOLS if not args.only or n in args.only.split(',')]
    tot = {'fwd': 0.0, 'bwd': 0.0}
    for name in names:
        N, T, H, W, C_, k, s, p = POOLS[name]
        g = torch.Generator(device='cpu').manual_seed(5)
        xa = ops.new_act(N, T, H, W, C_, dt, dev)
        xa.buf.copy_(torch.randn(xa.buf.shape, generator=g).clamp_(min=0).to(tdt))
        To, Ho, Wo = ops.conv_out_dims(xa, k, s, p)
        ya = ops.new_act(N, To, Ho, Wo, C_, dt, dev)
        dya = ops.new_act(N, To, Ho, Wo, C_, dt, dev)
        dya.buf.copy_(torch.randn(dya.buf.shape, generator=g).to(tdt))
        dxa = ops.new_act(N, T, H, W, C_, dt, dev)
        idx = torch.zeros(ya.rows, ops.cp8(C_), dtype=torch.uint8, device=dev)
        d = ops.pool_desc(dt, xa, ya, k, s, p)
        nin, nout = xa.rows * C_, ya.rows * C_
        ops.call('dv_maxpool3d_fwd', d, xa, ya, idx)
        for ps in args.passes.split(','):
            if ps == 'fwd':
                us = timed(lambda: ops.call('dv_maxpool3d_fwd', d, xa, ya, idx), args.reps)
                by = nin * es + nout * (es + 1)
            else:
                us = timed(lambda: ops.call('dv_maxpool3d_bwd', d, dya, idx, dxa, 0), args.reps)
                by = nout * (es + 1) + nin * es
            tot[ps] += us
            print('%-8s %s %-4s [%d,%d,%d,%d,%d] k%s s%s: %8.1f us  %7.1f GB/s algorithmic (%.1f MB)' %
                  (name, args.dtype, ps, N, T, H, W, C_, k, s, us, by / us / 1e3, by / 1e6))
    print('sum fwd %.1f us, bwd %.1f us' % (tot['fwd'], tot['bwd']))


if __name__ == '__main__':
    main()
