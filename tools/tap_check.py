#!/usr/bin/env python
"""Development aid for csrc/conv_tap.hip: correctness of the LDS-staged input-tile kernel against torch's conv3d on the CPU
(float64) for a list of shapes, forward / data gradient / `+=`, with the BatchNorm partials, plus which kernel the library
picked.  Run with DUALVAR_CONV_TAP_GRID=1 to force small problems onto it.

    DUALVAR_CONV_TAP_GRID=1 python tools/tap_check.py
"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualvar_amd import _lib as L          # noqa: E402
from dualvar_amd import ops                # noqa: E402

CASES = [
    # N, Cin, T, H, W, Cout, k, p
    (2, 16, 2, 9, 7, 64, (1, 3, 3), (0, 1, 1)),
    (3, 32, 3, 14, 14, 96, (1, 3, 3), (0, 1, 1)),
    (2, 64, 4, 28, 28, 192, (1, 3, 3), (0, 1, 1)),
    (1, 16, 2, 56, 56, 40, (1, 3, 3), (0, 1, 1)),
    (5, 48, 1, 5, 3, 24, (1, 3, 3), (0, 1, 1)),
    (3, 32, 4, 9, 7, 64, (3, 1, 1), (1, 0, 0)),
    (2, 192, 4, 14, 14, 192, (3, 1, 1), (1, 0, 0)),
    (5, 16, 2, 7, 7, 48, (3, 1, 1), (1, 0, 0)),
    (3, 32, 8, 6, 5, 80, (3, 1, 1), (1, 0, 0)),
    (7, 80, 4, 7, 5, 136, (3, 1, 1), (1, 0, 0)),
    (9, 208, 2, 7, 7, 208, (3, 1, 1), (1, 0, 0)),
    # more than 32 tiles per column tile: the two-level fold of the fused BatchNorm-backward reduce (49 tiles of 128 rows / 25 of 256;
    # R(2+1)D's conv3 block at two clips)
    (2, 128, 4, 28, 28, 230, (1, 3, 3), (0, 1, 1)),
    (8, 64, 2, 28, 28, 64, (3, 1, 1), (1, 0, 0)),
]


def check_fused_bn_reduce(lib, dev, dd_plain, dya, wd3, like, Ci, tag):
    """dv_conv3d_dgrad_bn_ws against the plain data gradient + the sums computed from it in float64: same dx bits, sums within
    fp32 rounding of a sum over all rows, identical bits from run to run, `+=` into the caller's sums, tickets left zero"""
    DT = L.DV_F32
    N, T, H, W, cip = like.N, like.T, like.H, like.W, like.cpitch
    dxp = ops.new_act(N, T, H, W, Ci, DT, dev, cpitch=cip, zero=True)
    ops.conv_dgrad(dd_plain, dya, wd3, dxp)
    g = torch.Generator().manual_seed(5)
    xb = ops.new_act(N, T, H, W, Ci, DT, dev, cpitch=cip, zero=True)
    xb.buf[:, :Ci] = (torch.randn(xb.rows, Ci, generator=g) + 0.2).to(dev)
    CPi = ops.cp8(Ci)

    def padded(t):
        o = torch.zeros(CPi, device=dev)
        o[:Ci] = t.to(dev)
        return o
    mean, invstd = padded(0.1 * torch.randn(Ci, generator=g)), padded(1 + 0.1 * torch.randn(Ci, generator=g).abs())
    scale, shift = padded(1 + 0.2 * torch.randn(Ci, generator=g)), padded(0.1 * torch.randn(Ci, generator=g))
    need = int(lib.dv_conv3d_dgrad_bn_workspace(C.byref(dd_plain)))
    assert need > 0, tag
    ws = torch.zeros(need // 4, device=dev)
    worst = 0.0
    for bflag in (0, L.DV_NO_RELU_MASK):
        gx, xx = dxp.buf[:, :CPi].double(), xb.buf[:, :CPi].double()
        act = (xb.buf[:, :CPi] * scale) + shift                    # two roundings, as the kernels compute it (no fma)
        gg = gx if bflag else torch.where(act > 0, gx, torch.zeros_like(gx))
        want = torch.stack([gg.sum(0), (gg * (xx - mean.double()) * invstd.double()).sum(0)])
        runs = []
        for rep in range(2):
            dxf = ops.new_act(N, T, H, W, Ci, DT, dev, cpitch=cip, zero=True)
            sums = torch.full((1, 2, CPi), 1.0, device=dev)            # += : starts at 1
            r = ops.bn_reduce_desc(xb, mean, invstd, scale, shift, sums, 1, bflag)
            L.check(lib.dv_conv3d_dgrad_bn_ws(C.byref(dd_plain), dya.ptr, wd3.data_ptr(), dxf.ptr, C.byref(r), ws.data_ptr(), need,
                                              ops.stream_ptr()), 'dv_conv3d_dgrad_bn_ws')
            torch.cuda.synchronize()
            assert torch.equal(dxf.buf, dxp.buf), tag + ': dx differs from the plain data gradient'
            runs.append(sums.clone())
        assert torch.equal(runs[0], runs[1]), tag + ': the fused sums are not reproducible'
        got = runs[0][0].double().cpu() - 1.0
        den = float(want.abs().max()) + float(gg.abs().sum(0).max()) * 1e-3
        e = float((got[:, :Ci] - want.cpu()[:, :Ci]).abs().max()) / float(gg.abs().sum(0).max())
        assert e == e
        worst = max(worst, e)
        assert float(got[:, Ci:].abs().max()) == 0.0 if CPi > Ci else True
    return worst


def check_bn_in(lib, dev, xa, dya, w, k, s, p, tag):
    """dv_conv3d_fwd_bn_in / dv_conv3d_wgrad_bn_in on the raw values of xa against dv_bn_apply + the plain entry points (which this
    script checks against float64): y, the BatchNorm partials and dW bit for bit, with and without the ReLU, scale / shift arrays
    only cp8(C) long, the channel pitch beyond them poisoned in x"""
    DT = L.DV_F32
    Ci, Co = xa.C, dya.C
    g = torch.Generator().manual_seed(7)
    CPi = ops.cp8(Ci)
    scale, shift = torch.zeros(CPi, device=dev), torch.zeros(CPi, device=dev)
    scale[:Ci] = (1 + 0.3 * torch.randn(Ci, generator=g)).to(dev)
    shift[:Ci] = (0.2 * torch.randn(Ci, generator=g)).to(dev)
    wp = ops.pack_weight(w.to(dev), xa.cpitch)
    w3 = ops.pack_w3(wp.view(Co, -1))
    d = ops.conv_desc(DT, xa, dya, k, s, p, flags=L.DV_STATS | L.DV_W3)
    dw_ = ops.conv_desc(DT, xa, dya, k, s, p)
    ok = int(lib.dv_conv3d_bn_in_ok(C.byref(d)))
    if not ok:
        return 0
    tiles = ops.stat_tiles(d)
    for relu in (True, False):
        act = ops.new_act(xa.N, xa.T, xa.H, xa.W, Ci, DT, dev, cpitch=xa.cpitch, zero=True)
        ops.call('dv_bn_apply', DT, xa, xa.ld, scale, shift, None, 0, act, act.ld, xa.rows, Ci, L.DV_RELU if relu else 0)
        y1, y2 = dya.like(), dya.like()
        y1.buf.zero_(); y2.buf.zero_()
        st1, st2 = torch.zeros(2, Co, tiles, device=dev), torch.zeros(2, Co, tiles, device=dev)
        bn = ops.bn_in_desc(scale, shift, relu)
        ops.conv_fwd(d, act, w3, None, y1, st1)
        ops.conv_fwd_bn_in(d, xa, bn, w3, y2, st2)
        dw1, dw2 = torch.zeros_like(wp), torch.zeros_like(wp)
        ops.conv_wgrad(dw_, act, dya, dw1)
        ops.conv_wgrad_bn_in(dw_, xa, bn, dya, dw2)
        torch.cuda.synchronize()
        assert torch.equal(y1.buf, y2.buf) and torch.equal(st1, st2), (tag, relu, float((y1.buf - y2.buf).abs().max()))
        assert torch.equal(dw1, dw2), (tag, relu, float((dw1 - dw2).abs().max()))
        assert float(y1.buf.abs().max()) > 0 and float(dw1.abs().max()) > 0 and bool(torch.isfinite(y2.buf).all())
    return ok


def check_pp_stem_forward(lib, dev, N, T, H, W, O, want_rows):
    """the pixel-pair stem form of the forward (conv_pp_fwd_kernel): dv_ingest_ncdhw_pad + the 1x7x4 / stride (1,2,1) conv over pixel
    pairs against torch's 7x7 / stride 2 / padding 3 conv3d in float64 (backbone/s3dg.py:151), with the BatchNorm partials and a
    sentinel behind the output"""
    DT = L.DV_F32
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, 3, T, H, W, generator=g)
    w = 0.2 * torch.randn(O, 3, 1, 7, 7, generator=g)
    yr = F.conv3d(x.double(), w.double(), None, (1, 2, 2), (0, 3, 3))
    a = ops.new_act(N, T, H + 6, W + 6, 3, DT, dev, cpitch=4, zero=True)
    ops.call('dv_ingest_ncdhw_pad', DT, x.to(dev), a, N, 3, T, H, W, 3 * T * H * W, 4, None, None, None, 0, 3)
    pairs = ops.Act(a.buf.view(-1, 8), N, T, H + 6, (W + 6) // 2, 8, 8, 0, DT, 8)
    To, Ho, Wo = yr.shape[2:]
    M = N * To * Ho * Wo
    OP = ops.cp8(O)
    ybuf = torch.full((M + 300, OP), 12345.0, dtype=torch.float32, device=dev)
    ybuf[:M] = 0
    y = ops.Act(ybuf[:M], N, To, Ho, Wo, O, OP, 0, DT, OP)
    d = ops.conv_desc(DT, pairs, y, (1, 7, 4), (1, 2, 1), (0, 0, 0), flags=L.DV_STATS | L.DV_W3)
    kind, rows = int(lib.dv_conv3d_tap_kind(C.byref(d), 0)), ops.tile_rows(d)
    assert kind == 3 and rows == want_rows, (kind, rows, want_rows)
    w8 = torch.zeros(O, 3, 1, 7, 8)
    w8[..., :7] = w
    wp = ops.pack_weight(w8, 4).to(dev)                       # [O][7*8 taps][4] == [O][7*4 pair taps][8]
    tiles = ops.stat_tiles(d)
    assert tiles * rows == M
    stats = torch.zeros(2, O, tiles, device=dev)
    ops.conv_fwd(d, pairs, ops.pack_w3(wp.view(O, -1)), None, y, stats)
    torch.cuda.synchronize()
    e_f = float((ops.act_to_ncdhw(y).double().cpu() - yr).abs().max() / yr.abs().max())
    local = torch.zeros(2 * O + 1, device=dev)
    ops.call('dv_bn_reduce_stats', stats, tiles, rows, O, M, O, local)
    mean_ref, var_ref = yr.mean(dim=(0, 2, 3, 4)), yr.var(dim=(0, 2, 3, 4), unbiased=False)
    e_m = float(((local[:O].cpu().double() / M) - mean_ref).abs().max() / var_ref.sqrt().max())
    e_v = float(((local[O:2 * O].cpu().double() / M) - var_ref).abs().max() / var_ref.max())
    assert bool((ybuf[M:] == 12345.0).all()), 'wrote behind the output'
    if OP > O:
        assert float(ybuf[:M, O:].abs().max()) == 0.0
    print('pixel-pair stem forward N%d T%d %dx%d Cout%d: tiles of %d rows | fwd %.2e mean %.2e var %.2e' % (N, T, H, W, O, rows, e_f, e_m, e_v), flush=True)
    assert e_f == e_f and e_f < 3e-6 and e_m < 3e-6 and e_v < 1e-5, (e_f, e_m, e_v)
    return e_f


def main():
    L.require_device()
    dev = torch.device('cuda:0')
    lib = L.load()
    DT = L.DV_F32
    worst = 0.0
    for (N, Ci, T, H, W, Co, k, p) in CASES:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(N, Ci, T, H, W, generator=g)
        w = torch.randn(Co, Ci, *k, generator=g) * (Ci * k[0] * k[1] * k[2]) ** -0.5
        gy = torch.randn(N, Co, T, H, W, generator=g)
        xr, wr = x.double().requires_grad_(True), w.double()
        yr = F.conv3d(xr, wr, None, 1, p)
        yr.backward(gy.double())
        xa = ops.act_from_ncdhw(x.to(dev), DT, cpitch=(Ci + 15) // 16 * 16)
        M_ = N * T * H * W
        ybuf = torch.full((M_ + 300, ops.cp8(Co)), 12345.0, dtype=torch.float32, device=dev)      # sentinel rows behind the output
        ybuf[:M_] = 0
        ya = ops.Act(ybuf[:M_], N, T, H, W, Co, ops.cp8(Co), 0, DT, ops.cp8(Co))
        d = ops.conv_desc(DT, xa, ya, k, (1, 1, 1), p, flags=L.DV_STATS | L.DV_W3)
        kind = lib.dv_conv3d_tap_kind(C.byref(d), 0)
        wp = ops.pack_weight(w.to(dev), xa.cpitch)
        tiles = ops.stat_tiles(d)
        stats = torch.zeros(2, Co, tiles, device=dev)
        ops.conv_fwd(d, xa, ops.pack_w3(wp.view(Co, -1)), None, ya, stats)
        torch.cuda.synchronize()
        e_f = float((ops.act_to_ncdhw(ya).double().cpu() - yr.detach()).abs().max() / yr.detach().abs().max())
        M = N * T * H * W
        local = torch.zeros(2 * Co + 1, device=dev)
        ops.call('dv_bn_reduce_stats', stats, tiles, ops.tile_rows(d), Co, M, Co, local)
        mean_ref = yr.detach().mean(dim=(0, 2, 3, 4))
        var_ref = yr.detach().var(dim=(0, 2, 3, 4), unbiased=False)
        e_m = float(((local[:Co].cpu().double() / M) - mean_ref).abs().max() / var_ref.sqrt().max())
        e_v = float(((local[Co:2 * Co].cpu().double() / M) - var_ref).abs().max() / var_ref.max())
        # data gradient (+ accumulate)
        cop = (Co + 15) // 16 * 16
        dya = ops.act_from_ncdhw(gy.to(dev), DT, cpitch=cop)
        taps = k[0] * k[1] * k[2]
        wd = torch.zeros(Ci, taps, cop, device=dev)
        wd[:, :, :Co] = w.to(dev).reshape(Co, Ci, taps).permute(1, 2, 0)
        xbuf = torch.full((M_ + 300, xa.cpitch), 12345.0, dtype=torch.float32, device=dev)
        xbuf[:M_] = 0
        dxa = ops.Act(xbuf[:M_], N, T, H, W, Ci, xa.cpitch, 0, DT, xa.cpitch)
        wd3 = ops.pack_w3(wd.view(Ci, -1))
        dd = ops.conv_desc(DT, dxa, dya, k, (1, 1, 1), p, flags=L.DV_W3)
        kind_d = lib.dv_conv3d_tap_kind(C.byref(dd), 1)
        ops.conv_dgrad(dd, dya, wd3, dxa)
        dd2 = ops.conv_desc(DT, dxa, dya, k, (1, 1, 1), p, flags=L.DV_W3 | L.DV_ACCUM)
        ops.conv_dgrad(dd2, dya, wd3, dxa)
        torch.cuda.synchronize()
        e_bn = check_fused_bn_reduce(lib, dev, dd, dya, wd3, dxa, Ci, 'N%d Cin%d k%s' % (N, Ci, k))
        e_d = float((ops.act_to_ncdhw(dxa).double().cpu() - 2 * xr.grad).abs().max() / (2 * xr.grad).abs().max())
        pad_ok = float(dxa.buf[:, Ci:].abs().max()) == 0.0 if xa.cpitch > Ci else True
        assert bool((ybuf[M_:] == 12345.0).all()) and bool((xbuf[M_:] == 12345.0).all()), 'wrote behind the output'
        if (k == (3, 1, 1) and T in (2, 4)) or k == (1, 3, 3):
            # the LDS-staged weight gradient (conv_tap_wgrad.hip): against float64, bit-reproducible, += , workspace content irrelevant
            wr = w.double().requires_grad_(True)
            F.conv3d(x.double(), wr, None, 1, p).backward(gy.double())
            dwd = ops.conv_desc(DT, xa, dya, k, (1, 1, 1), p)
            rr, cc_, ss = C.c_int32(), C.c_int32(), C.c_int32()
            assert lib.dv_conv3d_wgrad_tile(C.byref(dwd), C.byref(rr), C.byref(cc_), C.byref(ss)) == 0
            assert (rr.value, cc_.value) == (64, 192), (rr.value, cc_.value)
            need = ops.wgrad_workspace_bytes(dwd)
            runs = []
            for fill in (0, 0xFF):
                ws = torch.full((max(need, 16),), fill, dtype=torch.uint8, device=dev)
                dw = torch.zeros(Co, taps * xa.cpitch, device=dev)
                ops.conv_wgrad(dwd, xa, dya, dw, workspace=ws)
                runs.append(dw)
            torch.cuda.synchronize()
            assert torch.equal(runs[0], runs[1]), 'weight gradient depends on the workspace content'
            got = ops.unpack_weight(runs[0].view(Co, taps, xa.cpitch), w.shape).double().cpu()
            e_w = float((got - wr.grad).abs().max() / wr.grad.abs().max())
            if xa.cpitch > Ci:
                assert float(runs[0].view(Co, taps, xa.cpitch)[:, :, Ci:].abs().max()) == 0.0
            ops.conv_wgrad(dwd, xa, dya, runs[1], workspace=ws)
            assert torch.allclose(runs[1], 2 * runs[0], rtol=1e-6, atol=0)
            print('   weight gradient (LDS-staged, %d row splits): %.2e' % (ss.value, e_w), flush=True)
            assert e_w == e_w and e_w < 3e-6, e_w
            worst = max(worst, e_w)
        if k == (3, 1, 1):
            xp = ops.Act(xa.buf.clone(), N, T, H, W, Ci, xa.cpitch, 0, DT, xa.cpitch)
            xp.buf[:, Ci:] = 777.0                                 # pad lanes of the BatchNorm's input: must not reach the products
            okb = check_bn_in(lib, dev, xp, dya, w, k, (1, 1, 1), p, 'N%d Cin%d T%d' % (N, Ci, T))
            assert okb == (1 if T in (2, 4) else 0), (T, okb)
            print('   BatchNorm on load (fwd + weight gradient, bit for bit against apply-then-conv): %s' % ('ok' if okb else 'n/a (T = %d)' % T), flush=True)
        print('N%d Cin%d T%d %dx%d Cout%d k%s: rows %d: kind fwd %d dgrad %d | fwd %.2e mean %.2e var %.2e dgrad(+=) %.2e pad %s | fused bn sums %.2e' % (
            N, Ci, T, H, W, Co, 'x'.join(map(str, k)), ops.tile_rows(d), kind, kind_d, e_f, e_m, e_v, e_d, pad_ok, e_bn), flush=True)
        assert e_bn < 2e-6, e_bn
        assert all(e == e for e in (e_f, e_m, e_v, e_d)), 'NaN'
        worst = max(worst, e_f, e_m, e_v, e_d)
        assert pad_ok
    # t-strided kt x 1 x 1 data gradients (S3D-G's 7x1x1 / stride-2 stem conv and smaller kin): every parity class on the temporal
    # form (3 / 4 taps over the four frames of dY, rows scattered to every second frame of dX), `+=` included
    for (N, Ci, Ti, H, W, Co, kt, st, pt) in [(3, 32, 8, 9, 7, 64, 7, 2, 3), (2, 64, 8, 14, 14, 48, 7, 2, 3), (5, 16, 8, 6, 5, 32, 5, 2, 2)]:
        g = torch.Generator().manual_seed(2)
        k, sd, p = (kt, 1, 1), (st, 1, 1), (pt, 0, 0)
        x = torch.randn(N, Ci, Ti, H, W, generator=g)
        w = torch.randn(Co, Ci, *k, generator=g) * (Ci * kt) ** -0.5
        xr = x.double().requires_grad_(True)
        yr = F.conv3d(xr, w.double(), None, sd, p)
        To = yr.shape[2]
        gy = torch.randn(N, Co, To, H, W, generator=g)
        yr.backward(gy.double())
        cip, cop = (Ci + 15) // 16 * 16, (Co + 15) // 16 * 16
        dya = ops.act_from_ncdhw(gy.to(dev), DT, cpitch=cop)
        M_ = N * Ti * H * W
        xbuf = torch.full((M_ + 300, cip), 12345.0, dtype=torch.float32, device=dev)
        xbuf[:M_] = 0
        dxa = ops.Act(xbuf[:M_], N, Ti, H, W, Ci, cip, 0, DT, cip)
        wd = torch.zeros(Ci, kt, cop, device=dev)
        wd[:, :, :Co] = w.to(dev).reshape(Co, Ci, kt).permute(1, 2, 0)
        wd3 = ops.pack_w3(wd.view(Ci, -1))
        dd = ops.conv_desc(DT, dxa, dya, k, sd, p, flags=L.DV_W3)
        kind_d = lib.dv_conv3d_tap_kind(C.byref(dd), 1)
        assert kind_d == (2 if kt == 7 else 0), (kt, kind_d)      # (5 taps / stride 2: a class of two taps -- stays on conv_gemm)
        if kind_d == 2:
            ops.conv_dgrad(dd, dya, wd3, dxa)
            ops.conv_dgrad(ops.conv_desc(DT, dxa, dya, k, sd, p, flags=L.DV_W3 | L.DV_ACCUM), dya, wd3, dxa)
            torch.cuda.synchronize()
            e_d = float((ops.act_to_ncdhw(dxa).double().cpu() - 2 * xr.grad).abs().max() / (2 * xr.grad).abs().max())
            assert bool((xbuf[M_:] == 12345.0).all()), 'wrote behind the output'
            assert e_d == e_d
            worst = max(worst, e_d)
            e_bn = check_fused_bn_reduce(lib, dev, dd, dya, wd3, dxa, Ci, 'strided k%d' % kt)
            assert e_bn < 2e-6, e_bn
            # the stem form of the LDS-staged weight gradient (7 taps, stride 2, 8 -> 4 frames)
            wr = w.double().requires_grad_(True)
            F.conv3d(x.double(), wr, None, sd, p).backward(gy.double())
            xa = ops.act_from_ncdhw(x.to(dev), DT, cpitch=cip)
            dwd = ops.conv_desc(DT, xa, dya, k, sd, p)
            rr, cc_, ss = C.c_int32(), C.c_int32(), C.c_int32()
            assert lib.dv_conv3d_wgrad_tile(C.byref(dwd), C.byref(rr), C.byref(cc_), C.byref(ss)) == 0
            assert (rr.value, cc_.value) == (64, 7 * 32), (rr.value, cc_.value)
            need = ops.wgrad_workspace_bytes(dwd)
            runs = []
            for fill in (0, 0xFF):
                ws = torch.full((max(need, 16),), fill, dtype=torch.uint8, device=dev)
                dw = torch.zeros(Co, kt * cip, device=dev)
                ops.conv_wgrad(dwd, xa, dya, dw, workspace=ws)
                runs.append(dw)
            torch.cuda.synchronize()
            assert torch.equal(runs[0], runs[1])
            got = ops.unpack_weight(runs[0].view(Co, kt, cip), w.shape).double().cpu()
            e_w = float((got - wr.grad).abs().max() / wr.grad.abs().max())
            print('   weight gradient (LDS-staged stem form, %d row splits): %.2e' % (ss.value, e_w), flush=True)
            assert e_w == e_w and e_w < 3e-6, e_w
            worst = max(worst, e_w)
        else:
            e_d = float('nan')
        print('strided dgrad N%d Cin%d T%d->%d %dx%d Cout%d k%d s%d: kind %d | dgrad(+=) %.2e' % (N, Ci, Ti, To, H, W, Co, kt, st, kind_d, e_d), flush=True)
    # the pixel-pair stem form of the forward on small frames: a ragged last row block (99 rows), a full 256-row tile, 56-pixel lines
    for (N, T, H, W, O, rows) in [(2, 4, 18, 22, 24, 99), (1, 3, 32, 32, 64, 256), (2, 2, 24, 112, 40, 224), (1, 2, 10, 128, 72, 64)]:
        worst = max(worst, check_pp_stem_forward(lib, dev, N, T, H, W, O, rows))
    print('worst', worst)
    assert worst < 5e-6, worst
    print('ok')


if __name__ == '__main__':
    main()
