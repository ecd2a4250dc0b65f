#!/usr/bin/env python
"""Time dv_conv3d_fwd / dgrad / wgrad on single layers of the S3D-G pretrain step (bf16, random data).

    python tools/conv_microbench.py [--layers big|all] [--reps 20]

Prints one line per (layer, pass): microseconds, TFLOP/s, algorithmic GB/s.  A development aid for tile-shape and
pipeline experiments on the GPU box; not part of the product path or the tests."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualvar_amd import _lib as L          # noqa: E402
from dualvar_amd import ops                # noqa: E402

# (N, T, H, W, Cin, Cout, k, s, p) -- per-GPU batch of 128 clips of 8x112x112
LAYERS = {
    'stem2_7x1x1':    (128, 8, 56, 56, 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0)),
    'c2b_1x1x1':      (128, 4, 28, 28, 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    'c2c_1x3x3':      (128, 4, 28, 28, 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'c2c_3x1x1':      (128, 4, 28, 28, 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    'm3b_entry':      (128, 4, 14, 14, 192, 176, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    'm3b_1x3x3':      (128, 4, 14, 14, 96, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'm3c_entry':      (128, 4, 14, 14, 256, 288, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    'm3c_1x3x3':      (128, 4, 14, 14, 128, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'm3c_3x1x1':      (128, 4, 14, 14, 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    'm4b_entry':      (128, 2, 7, 7, 480, 304, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    'm4f_1x3x3':      (128, 2, 7, 7, 160, 320, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'm4f_3x1x1':      (128, 2, 7, 7, 320, 320, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    'm5c_1x3x3':      (128, 1, 3, 3, 192, 384, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'm5c_3x1x1':      (128, 1, 3, 3, 384, 384, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    # synthetic shapes for tile experiments (no padding waste with 128 / 256-wide tiles)
    'x256_3x1x1':     (128, 4, 14, 14, 256, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    'x128_1x3x3_28':  (128, 4, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
}
BIG = ['stem2_7x1x1', 'c2b_1x1x1', 'c2c_1x3x3', 'c2c_3x1x1', 'm3c_entry', 'm3c_1x3x3', 'm3c_3x1x1']


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
    ap.add_argument('--layers', default='big')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--passes', default='fwd,dgrad,wgrad')
    ap.add_argument('--no-stats', action='store_true', help='forward without the fused BatchNorm partials')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--w3', action='store_true', help='fp32: weights pre-split in fragment order (DV_W3), as the engine runs them')
    ap.add_argument('--bn-in', action='store_true', help='fwd / wgrad through dv_conv3d_*_bn_in on the RAW values (scale 1, shift 0, ReLU): the same operands as --data relu')
    ap.add_argument('--data', default='relu', choices=['relu', 'normal', 'zeros'], help='activation values (power, hence clocks, depend on them)')
    args = ap.parse_args()
    L.require_device()
    dev = torch.device('cuda:0')
    DT = L.DV_BF16 if args.dtype == 'bf16' else L.DV_F32
    tdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    es = 2 if args.dtype == 'bf16' else 4
    names = BIG if args.layers == 'big' else (list(LAYERS) if args.layers == 'all' else args.layers.split(','))
    for name in names:
        N, T, H, W, Ci, Co, k, s, p = LAYERS[name]
        x = ops.new_act(N, T, H, W, Ci, DT, dev)
        x.buf.normal_()                         # post-ReLU activations: half zeros, as inside the real step (clocks depend on it)
        if args.data == 'relu':
            x.buf.relu_()
        elif args.data == 'zeros':
            x.buf.zero_()
        To, Ho, Wo = ops.conv_out_dims(x, k, s, p)
        y = ops.new_act(N, To, Ho, Wo, Co, DT, dev)
        dy = y.like()
        dy.buf.normal_()
        if args.data == 'zeros':
            dy.buf.zero_()
        dx = x.like()
        taps = k[0] * k[1] * k[2]
        w = torch.randn(Co, taps * x.cpitch, device=dev).to(tdt)
        wd = torch.randn(Ci, taps * y.cpitch, device=dev).to(tdt)
        dw = torch.zeros(Co, taps * x.cpitch, device=dev)
        d = ops.conv_desc(DT, x, y, k, s, p, flags=0 if args.no_stats else L.DV_STATS)
        stats = torch.zeros(ops.stat_tiles(d) * 2 * Co, device=dev)
        dd = ops.conv_desc(DT, x, y, k, s, p)
        flops = 2.0 * y.rows * Co * Ci * taps
        bx, by = x.rows * x.cpitch * es, y.rows * y.cpitch * es
        ws = torch.empty(max(ops.wgrad_workspace_bytes(dd), 16), dtype=torch.uint8, device=dev)
        if args.w3:
            import ctypes as C
            d3 = ops.conv_desc(DT, x, y, k, s, p, flags=(0 if args.no_stats else L.DV_STATS) | L.DV_W3)
            dd3 = ops.conv_desc(DT, x, y, k, s, p, flags=L.DV_W3)
            w3, wd3 = ops.pack_w3(w.float().view(Co, -1)), ops.pack_w3(wd.float().view(Ci, -1))
            lib = L.load()
            fwd3 = lambda: L.check(lib.dv_conv3d_fwd(C.byref(d3), x.ptr, w3.data_ptr(), 0, y.ptr, stats.data_ptr(), ops.stream_ptr()), 'fwd3')
            dg3 = lambda: L.check(lib.dv_conv3d_dgrad(C.byref(dd3), dy.ptr, wd3.data_ptr(), dx.ptr, ops.stream_ptr()), 'dgrad3')
        jobs = {'fwd': (lambda: ops.conv_fwd(d, x, w, None, y, stats), bx + by),
                'dgrad': (lambda: ops.conv_dgrad(dd, dy, wd, dx), bx + by),
                'wgrad': (lambda: ops.conv_wgrad(dd, x, dy, dw, workspace=ws), bx + by)}
        if args.w3:
            jobs['fwd'] = (fwd3, bx + by)
            if max(s) == 1:
                jobs['dgrad'] = (dg3, bx + by)
        if args.bn_in:
            x.buf.normal_()
            sc, sh = torch.ones(x.cpitch, device=dev), torch.zeros(x.cpitch, device=dev)
            bn = ops.bn_in_desc(sc, sh, True)
            jobs['fwd'] = (lambda: ops.conv_fwd_bn_in(d3, x, bn, w3, y, stats), bx + by)
            jobs['wgrad'] = (lambda: ops.conv_wgrad_bn_in(dd, x, bn, dy, dw, workspace=ws), bx + by)
        for ps in args.passes.split(','):
            fn, nbytes = jobs[ps]
            us = timed(fn, args.reps)
            print('%-14s %-6s %8.1f us %7.1f TF %7.0f GB/s' % (name, ps, us, flops / us * 1e-6, nbytes / us * 1e-3), flush=True)


if __name__ == '__main__':
    main()
