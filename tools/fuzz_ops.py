#!/usr/bin/env python
"""Randomised shapes through the conv (fwd / dgrad / wgrad) and max-pool (fwd / bwd) kernels against PyTorch on the CPU
(development aid; the fixed cases live in tests/test_ops_gpu.py).    python tools/fuzz_ops.py [--n 60] [--seed 0]"""
import argparse
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualvar_amd import _lib as L          # noqa: E402
from dualvar_amd import ops                # noqa: E402

TOL = {L.DV_F32: 3e-5, L.DV_BF16: 2e-2}


def q(x, dt):
    return x.to(torch.bfloat16).float() if dt == L.DV_BF16 else x


def rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max() / (b.abs().max() + 1e-12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=60)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    L.require_device()
    dev = torch.device('cuda:0')
    rng = random.Random(args.seed)
    g = torch.Generator().manual_seed(args.seed)
    worst = {}
    for it in range(args.n):
        dt = rng.choice([L.DV_F32, L.DV_BF16])
        N, T, H, W = rng.randint(1, 3), rng.randint(1, 6), rng.randint(3, 17), rng.randint(3, 17)
        # ---- conv
        k = rng.choice([(1, 1, 1), (1, 3, 3), (3, 1, 1), (3, 3, 3), (5, 1, 1), (1, 5, 5)])
        s = tuple(rng.choice([1, 1, 2]) for _ in range(3))
        p = tuple(kk // 2 for kk in k)
        Cin, Cout = rng.choice([8, 16, 24, 40, 64, 83, 96]), rng.choice([8, 32, 48, 83, 128, 144])
        if all((d + 2 * pp - kk) // ss + 1 >= 1 for d, pp, kk, ss in zip((T, H, W), p, k, s)):
            x = q(torch.randn(N, Cin, T, H, W, generator=g), dt)
            w = q(torch.randn(Cout, Cin, *k, generator=g) * (Cin * k[0] * k[1] * k[2]) ** -0.5, dt)
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            yr = F.conv3d(xr, wr, None, s, p)
            gy = q(torch.randn(*yr.shape, generator=g), dt)
            yr.backward(gy)
            xa = ops.act_from_ncdhw(x.to(dev), dt)
            ya = ops.new_act(N, *yr.shape[2:], Cout, dt, dev, zero=True)
            d = ops.conv_desc(dt, xa, ya, k, s, p)
            wp = ops.pack_weight(w.to(dev), ops.cp8(Cin))
            ops.conv_fwd(d, xa, wp.to(ops.TORCH_DTYPE[dt]), None, ya, None)
            dya = ops.act_from_ncdhw(gy.to(dev), dt)
            dw = torch.zeros_like(wp)
            ops.conv_wgrad(d, xa, dya, dw)
            taps = k[0] * k[1] * k[2]
            wd = torch.zeros(Cin, taps, ops.cp8(Cout), device=dev)
            wd[:, :, :Cout] = w.to(dev).reshape(Cout, Cin, taps).permute(1, 2, 0)
            dxa = ops.new_act(N, T, H, W, Cin, dt, dev, zero=True)
            ops.conv_dgrad(d, dya, wd.to(ops.TORCH_DTYPE[dt]), dxa)
            errs = {'conv fwd': rel(ops.act_to_ncdhw(ya), yr.detach()), 'conv wgrad': rel(ops.unpack_weight(dw, w.shape), wr.grad),
                    'conv dgrad': rel(ops.act_to_ncdhw(dxa), xr.grad)}
            for name, e in errs.items():
                tag = (name, 'f32' if dt == L.DV_F32 else 'bf16')
                worst[tag] = max(worst.get(tag, 0.0), e)
                assert e <= TOL[dt] * (3 if 'grad' in name else 1), (name, e, dt, (N, Cin, T, H, W, Cout), k, s, p)
        # ---- pool
        k, s, p = rng.choice([((3, 3, 3), (1, 1, 1), (1, 1, 1)), ((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
                              ((2, 2, 2), (2, 2, 2), (0, 0, 0)), ((3, 3, 3), (1, 2, 2), (1, 1, 1))])
        C_ = rng.choice([8, 24, 40, 72, 132])
        if all((d + 2 * pp - kk) // ss + 1 >= 1 for d, pp, kk, ss in zip((T, H, W), p, k, s)):
            x = F.relu(q(torch.randn(N, C_, T, H, W, generator=g), dt))
            xr = x.clone().requires_grad_(True)
            yr = F.max_pool3d(xr, k, s, p)
            gy = q(torch.randn(*yr.shape, generator=g), dt)
            yr.backward(gy)
            xa = ops.act_from_ncdhw(x.to(dev), dt)
            ya = ops.new_act(N, *yr.shape[2:], C_, dt, dev)
            idx = torch.zeros(ya.rows, ops.cp8(C_), dtype=torch.uint8, device=dev)
            d = ops.pool_desc(dt, xa, ya, k, s, p)
            ops.call('dv_maxpool3d_fwd', d, xa, ya, idx)
            assert torch.equal(ops.act_to_ncdhw(ya).cpu(), yr.detach()), ('pool fwd', (N, C_, T, H, W), k, s, p)
            dya = ops.act_from_ncdhw(gy.to(dev), dt)
            dxa = ops.new_act(N, T, H, W, C_, dt, dev)
            ops.call('dv_maxpool3d_bwd', d, dya, idx, dxa, 0)
            e = rel(ops.act_to_ncdhw(dxa), q(xr.grad, dt))
            tag = ('pool bwd', 'f32' if dt == L.DV_F32 else 'bf16')
            worst[tag] = max(worst.get(tag, 0.0), e)
            assert e <= TOL[dt], ('pool bwd', e, (N, C_, T, H, W), k, s, p)
    # ---- per-sample reductions (self-gating mean / backward reduce): channel-chunked grids on the long-S levels
    for it in range(max(4, args.n // 6)):
        dt = rng.choice([L.DV_F32, L.DV_BF16])
        N, S, C_ = rng.randint(1, 140), rng.choice([9, 49, 130, 196, 784, 1000]), rng.choice([8, 40, 132, 256, 520])
        x = q(torch.randn(N * S, C_, generator=g), dt)
        dy = q(torch.randn(N * S, C_, generator=g), dt)
        gate = torch.rand(N, C_, generator=g)
        xa = ops.new_act(N, 1, 1, S, C_, dt, dev)
        xa.buf[:, :C_] = x.to(dev).to(ops.TORCH_DTYPE[dt])
        dya = ops.new_act(N, 1, 1, S, C_, dt, dev)
        dya.buf[:, :C_] = dy.to(dev).to(ops.TORCH_DTYPE[dt])
        mean = torch.full((N, C_), 9.0, device=dev)
        ops.call('dv_spatial_mean', dt, xa, xa.ld, N, S, C_, mean)
        e = rel(mean, x.view(N, S, C_).double().mean(1).float())
        worst[('spatial mean', 'f32' if dt == L.DV_F32 else 'bf16')] = max(worst.get(('spatial mean', 'f32' if dt == L.DV_F32 else 'bf16'), 0.0), e)
        assert e <= 2e-5, ('spatial_mean', e, N, S, C_)
        dpre = torch.full((N, C_), 9.0, device=dev)
        ops.call('dv_gate_bwd_reduce', dt, dya, dya.ld, xa, xa.ld, gate.to(dev), N, S, C_, dpre, 0)
        want = ((dy.double() * x.double()).view(N, S, C_).sum(1) * (gate * (1 - gate)).double()).float()
        e = rel(dpre, want)
        worst[('gate reduce', 'f32' if dt == L.DV_F32 else 'bf16')] = max(worst.get(('gate reduce', 'f32' if dt == L.DV_F32 else 'bf16'), 0.0), e)
        assert e <= 5e-5, ('gate_bwd_reduce', e, N, S, C_)
    for k_, v in sorted(worst.items()):
        print('%-12s %-5s worst rel err %.2e' % (k_[0], k_[1], v))
    print('ok: %d random cases' % args.n)


if __name__ == '__main__':
    main()
