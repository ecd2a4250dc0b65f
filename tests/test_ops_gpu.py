"""GPU parity of every HIP kernel against the fp32 PyTorch-CPU op the oracle is made of.

Tolerances: DV_F32 path -> 2e-5 relative to the tensor's max (fp32 storage, statistics and accumulation; products as six
exact bf16 partial products on the matrix cores -- error at fp32 rounding level -- or, under DUALVAR_F32_EXACT=1, the
f32-input MFMA); DV_BF16 path -> inputs are rounded to bf16 first, outputs compared at 1.5e-2 of the tensor's max
(bf16 storage rounding, fp32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from dualvar_amd import ops  # noqa: E402
from dualvar_amd.ops import DV_BF16, DV_F32  # noqa: E402

TOL = {DV_F32: 2e-5, DV_BF16: 1.5e-2}
from dualvar_amd import _lib as _L  # noqa: E402
_EXACT = _L.f32_exact()          # A/B runs of the whole suite on the exact-f32 MFMA kernels


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(x, dtype):
    return x.to(torch.bfloat16).float() if dtype == DV_BF16 else x


def close(got, ref, dtype, what, factor=1.0):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    den = ref.abs().max().item() + 1e-12
    assert err <= TOL[dtype] * factor * den + 1e-7, f'{what}: max err {err:.3e} vs max |ref| {den:.3e}'


CONV_CASES = [
    # name, N, Cin, T, H, W, Cout, k, s, p
    ('stem_sp7', 2, 3, 4, 30, 30, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
    ('stem_tm7', 2, 64, 8, 9, 9, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0)),
    ('pw', 3, 64, 2, 7, 7, 96, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('sp3', 2, 96, 2, 7, 9, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('tm3', 2, 32, 4, 5, 5, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('small_c', 2, 16, 2, 6, 6, 48, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('c24', 2, 24, 2, 7, 7, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r21d_odd', 1, 64, 2, 8, 8, 83, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ('r21d_odd_in', 1, 83, 4, 6, 6, 64, (3, 1, 1), (2, 1, 1), (1, 0, 0)),
    ('pw_s', 2, 64, 4, 8, 8, 42, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ('pw_t', 2, 42, 4, 4, 4, 128, (1, 1, 1), (2, 1, 1), (0, 0, 0)),
    ('full3', 2, 32, 4, 8, 8, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ('big_n', 2, 160, 1, 3, 3, 320, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r50_stem', 1, 3, 8, 20, 20, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3)),
    # windows with taps that are padding for EVERY row (trim_dead_taps in csrc/conv.hip): one-frame maps under 3x1x1 / 3x3x3
    ('tm3_t1', 3, 64, 1, 3, 3, 96, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('tm3_t1_c24', 2, 24, 1, 4, 4, 48, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('full3_t1', 2, 32, 1, 2, 2, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('full3_t1_h1', 4, 48, 1, 1, 5, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # the pixel-pair stem's geometry: 8 channels (= half a K tile of 16 f32), even kernel width, padding-free, stride (1, 2, 1):
    # (generic gather: a K tile spans two taps)
    # the few-row K-split kernel (conv_gemm_ks): its 64-column form (more than 256 tiles of 32 columns), and together with a
    # trimmed window (Mixed_5c's 3x1x1 on one-frame maps: weights addressed at the live tap)
    ('ks64', 4, 64, 4, 20, 20, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('tm3_t1_ks', 4, 384, 1, 3, 3, 384, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('pair_stem', 2, 8, 4, 20, 22, 64, (1, 7, 4), (1, 2, 1), (0, 0, 0)),
    ('pair_stem_t', 1, 8, 9, 12, 10, 64, (5, 7, 4), (2, 2, 1), (2, 0, 0)),
]


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
@pytest.mark.parametrize('case', [('sp3_wide', 8, 32, 4, 16, 16, 48, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                                  ('tm7_s2', 4, 64, 8, 12, 12, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0)),
                                  ('pw_192', 4, 192, 4, 14, 14, 176, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
                                  ('full3_208', 6, 96, 4, 10, 10, 208, (3, 3, 3), (1, 1, 1), (1, 1, 1))],
                         ids=lambda c: c[0])
def test_conv_wgrad_row_splits_are_deterministic(gpu, dtype, case):
    """The weight gradient over MANY row splits (scratch partial tiles + ordered reduce, include/dualvar_hip.h): equal to
    torch's, bit-for-bit reproducible from run to run, `+=` into the gradient arena, and independent of what the
    (shared) workspace held before."""
    name, N, Cin, T, H, W, Cout, k, s, p = case
    x = q(rnd(N, Cin, T, H, W, seed=1), dtype)
    w = q(rnd(Cout, Cin, *k, seed=2, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5), dtype)
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(x, wr, None, s, p)
    gy = q(rnd(*yr.shape, seed=3), dtype)
    yr.backward(gy)
    xa = ops.act_from_ncdhw(x.to(gpu), dtype)
    dya = ops.act_from_ncdhw(gy.to(gpu), dtype)
    d = ops.conv_desc(dtype, xa, dya, k, s, p)
    need = ops.wgrad_workspace_bytes(d)
    assert need > 0, 'case too small to split: %s' % name
    wp = ops.pack_weight(w.to(gpu), ops.cp8(Cin))
    ws = torch.empty(need, dtype=torch.uint8, device=gpu)
    runs = []
    for fill in (0, 0xFF, 0x7F):                       # 0xFF.. = NaN patterns: every slab word that is read was written
        ws.fill_(fill)
        dw = torch.zeros_like(wp)
        ops.conv_wgrad(d, xa, dya, dw, workspace=ws)
        runs.append(dw)
    torch.cuda.synchronize()
    close(ops.unpack_weight(runs[0], w.shape), wr.grad, dtype, name + ' wgrad')
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    ops.conv_wgrad(d, xa, dya, runs[2], workspace=ws)     # accumulates
    assert torch.allclose(runs[2], 2 * runs[0], rtol=1e-6, atol=0)
    with pytest.raises(Exception):                         # a workspace that is too small is rejected, nothing is launched
        ops.conv_wgrad(d, xa, dya, dw, workspace=ws[:need // 2])


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(gpu, dtype, case):
    name, N, Cin, T, H, W, Cout, k, s, p = case
    x = q(rnd(N, Cin, T, H, W, seed=1), dtype)
    w = q(rnd(Cout, Cin, *k, seed=2, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5), dtype)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, s, p)
    gy = q(rnd(*yr.shape, seed=3), dtype)
    yr.backward(gy)

    cin_pitch = 4 if Cin == 3 else ops.cp8(Cin)
    xa = ops.act_from_ncdhw(x.to(gpu), dtype, cpitch=cin_pitch)
    To, Ho, Wo = yr.shape[2:]
    # write y into a channel slice of a wider buffer (concat-by-slice)
    wide = ops.new_act(N, To, Ho, Wo, ops.cp8(Cout) + 16, dtype, gpu, zero=True)
    ya = wide.slice(8, Cout)
    d = ops.conv_desc(dtype, xa, ya, k, s, p, flags=ops.DV_STATS)
    wp = ops.pack_weight(w.to(gpu), cin_pitch)
    wpc = wp.to(ops.TORCH_DTYPE[dtype])
    tiles = ops.stat_tiles(d)
    stats = torch.zeros(2, Cout, tiles, device=gpu)            # [2][Cout][tiles]
    ops.conv_fwd(d, xa, wpc, None, ya, stats)
    torch.cuda.synchronize()
    close(ops.act_to_ncdhw(ya), yr, dtype, name + ' fwd')
    # neighbours of the slice untouched, pad lanes zero
    assert float(wide.buf[:, :8].abs().max()) == 0.0
    if ops.cp8(Cout) > Cout:
        assert float(wide.buf[:, 8 + Cout:8 + ops.cp8(Cout)].abs().max()) == 0.0

    # statistics: sum and M2 recombine to the batch mean / biased variance of the stored values
    M = N * To * Ho * Wo
    local = torch.zeros(2 * Cout + 1, device=gpu)
    ops.call('dv_bn_reduce_stats', stats, tiles, ops.tile_rows(d), Cout, M, Cout, local)
    ys = ops.act_to_ncdhw(ya)
    mean_ref = ys.mean(dim=(0, 2, 3, 4)).cpu()
    var_ref = ys.var(dim=(0, 2, 3, 4), unbiased=False).cpu()
    close(local[:Cout] / M, mean_ref, DV_F32, name + ' mean', factor=5)
    close(local[Cout:2 * Cout] / M, var_ref, DV_F32, name + ' var', factor=20)
    assert float(local[2 * Cout]) == M

    if dtype == DV_F32 and not _EXACT:          # (DUALVAR_F32_EXACT=1: the exact-f32 kernels take no pre-split weights)
        # the same forward with the weights handed over pre-split in fragment order (DV_W3, dv_pack_w3): the engine's fp32 path
        from dualvar_amd._lib import DV_W3
        wide3 = ops.new_act(N, To, Ho, Wo, ops.cp8(Cout) + 16, dtype, gpu, zero=True)
        y3 = wide3.slice(8, Cout)
        d3 = ops.conv_desc(dtype, xa, y3, k, s, p, flags=ops.DV_STATS | DV_W3)
        if name in ('ks64', 'tm3_t1_ks', 'big_n', 'sp3'):
            import ctypes as C
            want = 64 if name == 'ks64' else 32
            assert _L.load().dv_conv3d_ksplit_cols(C.byref(d3), 0) == want, name
        tiles3 = ops.stat_tiles(d3)              # (the pre-split path may pick another kernel, hence another partial tiling)
        stats3 = torch.zeros(2, Cout, tiles3, device=gpu)
        ops.conv_fwd(d3, xa, ops.pack_w3(wp.view(Cout, -1)), None, y3, stats3)
        torch.cuda.synchronize()
        close(ops.act_to_ncdhw(y3), yr, dtype, name + ' fwd (pre-split weights)')
        assert float((ops.act_to_ncdhw(y3) - ops.act_to_ncdhw(ya)).abs().max()) <= 2e-6 * float(yr.abs().max())
        if tiles3 == tiles and ops.tile_rows(d3) == ops.tile_rows(d):
            assert torch.allclose(stats3, stats, rtol=1e-4, atol=1e-5)
        local3 = torch.zeros(2 * Cout + 1, device=gpu)
        ops.call('dv_bn_reduce_stats', stats3, tiles3, ops.tile_rows(d3), Cout, M, Cout, local3)
        close(local3[:Cout] / M, mean_ref, DV_F32, name + ' mean (pre-split weights)', factor=5)
        close(local3[Cout:2 * Cout] / M, var_ref, DV_F32, name + ' var (pre-split weights)', factor=20)

    # wgrad
    dya = ops.act_from_ncdhw(gy.to(gpu), dtype)
    d2 = ops.conv_desc(dtype, xa, dya, k, s, p)
    dw = torch.zeros_like(wp)
    ops.conv_wgrad(d2, xa, dya, dw)
    close(ops.unpack_weight(dw, w.shape), wr.grad, dtype, name + ' wgrad')
    if cin_pitch > Cin:
        assert float(dw[:, :, Cin:].abs().max()) == 0.0

    # dgrad (never needed for the RGB input)
    if Cin != 3:
        taps = k[0] * k[1] * k[2]
        cout_pitch = ops.cp8(Cout)
        wd = torch.zeros(Cin, taps, cout_pitch, device=gpu)
        wd[:, :, :Cout] = w.to(gpu).reshape(Cout, Cin, taps).permute(1, 2, 0)
        wd = wd.to(ops.TORCH_DTYPE[dtype])
        dxa = ops.new_act(N, T, H, W, Cin, dtype, gpu, zero=True)
        ops.conv_dgrad(d2, dya, wd, dxa)
        close(ops.act_to_ncdhw(dxa), xr.grad, dtype, name + ' dgrad')
        # accumulate flag
        d3 = ops.conv_desc(dtype, xa, dya, k, s, p, flags=ops.DV_ACCUM)
        ops.conv_dgrad(d3, dya, wd, dxa)
        close(ops.act_to_ncdhw(dxa), 2 * xr.grad, dtype, name + ' dgrad accum', factor=2)
        if dtype == DV_F32 and max(s) == 1 and not _EXACT:
            from dualvar_amd._lib import DV_W3
            dx3 = ops.new_act(N, T, H, W, Cin, dtype, gpu, zero=True)
            ops.conv_dgrad(ops.conv_desc(dtype, xa, dya, k, s, p, flags=DV_W3), dya, ops.pack_w3(wd.view(Cin, -1)), dx3)
            close(ops.act_to_ncdhw(dx3), xr.grad, dtype, name + ' dgrad (pre-split weights)')
        # the data gradient with the BatchNorm-backward reduce of the layer in front fused into its epilogue
        # (dv_conv3d_dgrad_bn): dx bit-identical to the plain call, sums == dv_bn_bwd_reduce(DV_MASK_FROM_X) on that dx
        from dualvar_amd._lib import DV_NO_RELU_MASK
        CPi = ops.cp8(Cin)
        xb = ops.act_from_ncdhw(q(rnd(N, Cin, T, H, W, seed=5) + 0.2, dtype).to(gpu), dtype)     # the BatchNorm's input

        def padded(t):
            o = torch.zeros(CPi, device=gpu)
            o[:Cin] = t.to(gpu)
            return o
        mean, invstd = padded(0.1 * rnd(Cin, seed=6)), padded(1 + 0.1 * rnd(Cin, seed=7).abs())
        scale, shift = padded(1 + 0.2 * rnd(Cin, seed=8)), padded(0.1 * rnd(Cin, seed=9))
        for bflag in (0, DV_NO_RELU_MASK):
            dxp = ops.new_act(N, T, H, W, Cin, dtype, gpu, zero=True)
            ops.conv_dgrad(d2, dya, wd, dxp)
            gx, xx = dxp.buf[:, :CPi].float(), xb.buf[:, :CPi].float()
            act = xx * scale                                     # two roundings, as the kernels compute it (no fma)
            act = act + shift
            gg = gx if bflag else torch.where(act > 0, gx, torch.zeros_like(gx))
            want = torch.stack([gg.double().sum(0), (gg * (xx - mean) * invstd).double().sum(0)]).float()[None]
            dxf = ops.new_act(N, T, H, W, Cin, dtype, gpu, zero=True)
            sums = torch.zeros(3, 2, CPi, device=gpu)
            ops.conv_dgrad_bn(d2, dya, wd, dxf, ops.bn_reduce_desc(xb, mean, invstd, scale, shift, sums, 3, bflag))
            assert torch.equal(dxf.buf, dxp.buf)
            got = sums.sum(0)
            tol = (2e-5 if dtype == DV_F32 else 2e-5) * float(want.abs().max()) + 1e-6
            assert float((got - want[0]).abs().max()) <= tol, (name, float((got - want[0]).abs().max()), float(want.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('shape', [
    # (N, Cin, T, H, W, Cout, k, s, p): the RGB-stem form (8-channel pixel pairs, few input channels, many rows), a ragged row
    # count (last 32-row step and last split partial), and two channel tiles with a ragged last one
    (4, 8, 4, 30, 26, 64, (1, 7, 4), (1, 2, 1), (0, 0, 0)),
    (3, 8, 3, 17, 13, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 16, 4, 20, 20, 88, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    (8, 8, 8, 64, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),         # 262 144 rows: 512 workgroups, two per CU, several row-table rounds each
    (4, 8, 8, 64, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
])
def test_weight_gradient_with_the_batchnorm_backward_apply_inside(gpu, shape, relu):
    """dv_conv3d_wgrad_bn (first conv of a network: backbone/s3dg.py:151 -> BatchNorm -> ReLU) against the two launches it
    replaces, dv_bn_bwd_apply + dv_conv3d_wgrad: the operand is formed with the same expression, so dW, dgamma and dbeta
    must agree bit for bit; and against torch autograd of conv -> batch_norm -> relu with the fp32 tolerance."""
    import ctypes as C
    from dualvar_amd import _lib as L_
    from dualvar_amd._lib import DV_NO_RELU_MASK, DV_MASK_FROM_X
    if _EXACT:
        pytest.skip('the fused form rides on the split-mode kernel')
    N, Cin, T, H, W, Cout, k, s, p = shape
    dtype = DV_F32
    lib = L_.load()
    x = rnd(N, Cin, T, H, W, seed=1)
    w = 0.2 * rnd(Cout, Cin, *k, seed=2)
    gamma, beta = 1 + 0.3 * rnd(Cout, seed=3), 0.2 * rnd(Cout, seed=4)
    # reference in float64 (the fp32 CPU autograd's own rounding over 1e5 cancelling rows is larger than the tolerance below)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yc = F.conv3d(xr, wr, None, s, p)
    yb = F.batch_norm(yc, None, None, gr, br, True, 0.1, 1e-5)
    out = F.relu(yb) if relu else yb
    gy = rnd(*out.shape, seed=5)
    out.backward(gy.double())

    xa = ops.act_from_ncdhw(x.to(gpu), dtype)
    To, Ho, Wo = out.shape[2:]
    ya = ops.new_act(N, To, Ho, Wo, Cout, dtype, gpu, zero=True)
    d = ops.conv_desc(dtype, xa, ya, k, s, p, flags=ops.DV_STATS)
    wp = ops.pack_weight(w.to(gpu), xa.cpitch)
    M, CP = ya.rows, ops.cp8(Cout)
    tiles = ops.stat_tiles(d)
    stats = torch.zeros(2, Cout, tiles, device=gpu)
    ops.conv_fwd(d, xa, wp, None, ya, stats)

    def padded(t):
        o = torch.zeros(CP, device=gpu)
        o[:Cout] = t.to(gpu)
        return o
    gam, bet = padded(gamma), padded(beta)
    mean, invstd, scale, shift = (torch.zeros(CP, device=gpu) for _ in range(4))
    local = torch.zeros(2 * Cout + 1, device=gpu)
    rm, rv = torch.zeros(Cout, device=gpu), torch.ones(Cout, device=gpu)
    ops.call('dv_bn_stats_finalize', stats, tiles, ops.tile_rows(d), Cout, M, Cout, local, gam, bet, 1e-5, 0.1, rm, rv,
             mean, invstd, scale, shift)
    ga = ops.act_from_ncdhw(gy.to(gpu), dtype)                  # dL/d(relu output)
    flags = (DV_MASK_FROM_X if relu else DV_NO_RELU_MASK)
    nrep = 3
    sums = torch.zeros(nrep, 2, CP, device=gpu)
    if not relu:
        ops.call('dv_bn_bwd_reduce', dtype, ga, ga.ld, ya, ya.ld, ya, ya.ld, mean, invstd, M, Cout, DV_NO_RELU_MASK, sums, nrep, None)
    else:          # (the mask-from-x reduce only exists as the multi-tensor entry: any sums will do for this comparison)
        xx = ya.buf[:, :Cout]
        act = xx * scale[:Cout]
        act = act + shift[:Cout]
        gg = torch.where(act > 0, ga.buf[:, :Cout], torch.zeros_like(xx))
        sums[0, 0, :Cout] = gg.double().sum(0).float()
        sums[0, 1, :Cout] = (gg * (xx - mean[:Cout]) * invstd[:Cout]).double().sum(0).float()
    # reference: apply, then the plain weight gradient
    dxa = ops.new_act(N, To, Ho, Wo, Cout, dtype, gpu, zero=True)
    dg0, db0 = torch.zeros(Cout, device=gpu), torch.zeros(Cout, device=gpu)
    if relu:
        it = (L_.BnItem * 1)()
        i0 = it[0]
        i0.x, i0.ldx, i0.y, i0.ldy = ya.ptr, ya.ld, ya.ptr, ya.ld
        i0.dy, i0.lddy, i0.dx, i0.lddx = ga.ptr, ga.ld, dxa.ptr, dxa.ld
        i0.mean, i0.invstd, i0.scale, i0.shift = (t.data_ptr() for t in (mean, invstd, scale, shift))
        i0.gamma, i0.sums, i0.n_rep = gam.data_ptr(), sums.data_ptr(), nrep
        i0.dgamma, i0.dbeta = dg0.data_ptr(), db0.data_ptr()
        i0.inv_count, i0.dparam_scale = 1.0 / M, 1.0
        i0.M, i0.C, i0.bwd_flags = M, Cout, flags
        nb = max(1, min(2048, (M * (CP // 4) + 255) // 256))
        i0.blk_bapply = nb
        tab = torch.frombuffer(bytearray(bytes(it)), dtype=torch.uint8).to(gpu)
        ops.call('dv_bn_bwd_apply_multi', dtype, tab.data_ptr(), 1, nb, Cout)
    else:
        ops.call('dv_bn_bwd_apply', dtype, ga, ga.ld, ya, ya.ld, ya, ya.ld, mean, invstd, gam, sums, nrep, 1.0 / M, 1.0,
                 dg0, db0, dxa, dxa.ld, None, 0, M, Cout, flags)
    d2 = ops.conv_desc(dtype, xa, dxa, k, s, p)
    assert lib.dv_conv3d_wgrad_bn_ok(C.byref(d2)) == 1
    dw0 = torch.zeros(Cout, k[0] * k[1] * k[2], xa.cpitch, device=gpu)
    ops.conv_wgrad(d2, xa, dxa, dw0)
    # fused
    dw1 = torch.zeros_like(dw0)
    dg1, db1 = torch.zeros(Cout, device=gpu), torch.zeros(Cout, device=gpu)
    r = L_.BnBwd()
    r.x, r.ldx = ya.ptr, ya.ld
    r.mean, r.invstd, r.gamma, r.scale, r.shift = (t.data_ptr() for t in (mean, invstd, gam, scale, shift))
    r.sums, r.n_rep, r.flags = sums.data_ptr(), nrep, (0 if relu else DV_NO_RELU_MASK)
    r.dgamma, r.dbeta, r.inv_count, r.dparam_scale = dg1.data_ptr(), db1.data_ptr(), 1.0 / M, 1.0
    need = ops.wgrad_workspace_bytes(d2)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=gpu)
    L_.check(lib.dv_conv3d_wgrad_bn(C.byref(d2), xa.ptr, ga.ptr, dw1.data_ptr(), ws.data_ptr(), need, C.byref(r),
                                    ops.stream_ptr()), 'dv_conv3d_wgrad_bn')
    torch.cuda.synchronize()
    # ... and run-to-run identical.  (Many times on the large shape: its grid puts two workgroups on every CU, the case in which
    # an earlier code generation of this kernel -- not this one -- differed from run to run in the last bits; DESIGN.md.)
    r.dgamma = r.dbeta = None
    for _ in range(40 if M > 100000 else 2):
        dw2 = torch.zeros_like(dw0)
        L_.check(lib.dv_conv3d_wgrad_bn(C.byref(d2), xa.ptr, ga.ptr, dw2.data_ptr(), ws.data_ptr(), need, C.byref(r),
                                        ops.stream_ptr()), 'dv_conv3d_wgrad_bn')
        torch.cuda.synchronize()
        assert torch.equal(dw1, dw2), float((dw1 - dw2).abs().max())
    assert torch.equal(dw1, dw0), float((dw1 - dw0).abs().max())
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0)
    close(ops.unpack_weight(dw1, w.shape), wr.grad, dtype, 'fused wgrad vs autograd', factor=20)
    close(dg1, gr.grad, dtype, 'dgamma', factor=20)
    close(db1, br.grad, dtype, 'dbeta', factor=20)


def test_pack_dgrad_and_cast(gpu):
    import ctypes as C
    from dualvar_amd import _lib as L
    shapes = [(64, 32, 9), (83, 64, 3), (48, 16, 1)]
    master, descs, bmap, soff, doff = [], [], [], 0, 0
    for i, (O, I, taps) in enumerate(shapes):
        cinp, coutp = ops.cp8(I), ops.cp8(O)
        w = rnd(O, taps, cinp, seed=10 + i)
        w[:, :, I:] = 0
        master.append(w.reshape(-1))
        descs.append((soff, doff, O, I, taps, cinp, coutp))
        bmap += [(i, c) for c in range(I)]
        soff += w.numel()
        doff += I * taps * coutp
    m = torch.cat(master).to(gpu)
    darr = (L.PackDesc * len(descs))()
    for j, t in enumerate(descs):
        darr[j].src_off, darr[j].dst_off, darr[j].Cout, darr[j].Cin, darr[j].taps, darr[j].cin_pitch, darr[j].cout_pitch = t
    dbytes = torch.frombuffer(bytearray(bytes(darr)), dtype=torch.uint8).to(gpu)
    bm = torch.tensor(bmap, dtype=torch.int32).to(gpu)
    for dtype in (DV_F32, DV_BF16):
        dst = torch.full((doff,), 7.0, dtype=ops.TORCH_DTYPE[dtype], device=gpu)
        ops.call('dv_pack_dgrad_weights', dtype, m, dst, dbytes, bm, len(bmap))
        for (so, do, O, I, taps, cinp, coutp), w in zip(descs, master):
            ref = torch.zeros(I, taps, coutp)
            ref[:, :, :O] = w.reshape(O, taps, cinp)[:, :, :I].permute(2, 1, 0)
            close(dst[do:do + I * taps * coutp].reshape(I, taps, coutp), q(ref, dtype), DV_F32, 'pack')
        c = torch.empty(m.numel(), dtype=ops.TORCH_DTYPE[dtype], device=gpu)
        ops.call('dv_cast_arena', dtype, m, c, m.numel())
        close(c, q(m.cpu(), dtype), DV_F32, 'cast')


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
def test_ingest(gpu, dtype):
    B, V, T, H, W = 3, 2, 4, 6, 5
    block = rnd(B, V, 3, T, H, W, seed=4)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    x = block.reshape(B * V, 3, T, H, W).to(gpu)
    a = ops.new_act(B * V, T, H, W, 3, dtype, gpu, cpitch=4)
    ops.call('dv_ingest_ncdhw', dtype, x, a, B * V, 3, T, H, W, 3 * T * H * W, 4, mean.to(gpu), (1 / std).to(gpu), None, 0)
    ref = (block.reshape(B * V, 3, T, H, W) - mean.view(1, 3, 1, 1, 1)) / std.view(1, 3, 1, 1, 1)
    close(ops.act_to_ncdhw(a), q(ref, dtype), dtype, 'ingest', factor=0.5 if dtype == DV_BF16 else 1)
    assert float(a.buf[:, 3].abs().max()) == 0.0
    # strided view (one view of the block) + segment permutation (simclr.py:378-383)
    perm = torch.tensor([[1, 0], [0, 1], [1, 0]], dtype=torch.int32)
    v2 = block.to(gpu)[:, 1]
    a2 = ops.new_act(B, T, H, W, 3, dtype, gpu, cpitch=4)
    ops.call('dv_ingest_ncdhw', dtype, v2, a2, B, 3, T, H, W, V * 3 * T * H * W, 4, None, None, perm.to(gpu), 2)
    xv = block[:, 1].reshape(B, 3, 2, T // 2, H, W)
    ref2 = torch.gather(xv, 2, perm.long().view(B, 1, 2, 1, 1, 1).expand_as(xv)).reshape(B, 3, T, H, W)
    close(ops.act_to_ncdhw(a2), q(ref2, dtype), dtype, 'ingest perm', factor=0.5 if dtype == DV_BF16 else 1)


def _augment(gpu, dtype, frames, table, N, T, H, W, mean, std, perm=None, pad=0, blur=None):
    from dualvar_amd.utils.transforms import AUG_BLUR, AUG_ROW
    assert table.dtype == AUG_ROW
    if blur is not None:
        assert blur.dtype == AUG_BLUR and len(blur) == len(table)
        bl = torch.from_numpy(blur.view(np.uint8).copy()).to(gpu)
        btmp = torch.empty(N * T * H * W * 3, dtype=torch.uint8, device=gpu)
    a = ops.new_act(N, T, H + 2 * pad, W + 2 * pad, 3, dtype, gpu, cpitch=4, zero=True)
    fr = torch.from_numpy(np.ascontiguousarray(frames)).to(gpu)
    tb = torch.from_numpy(table.view(np.uint8).copy()).to(gpu)
    scratch = torch.full((N * T,), float('nan'), device=gpu)
    ops.call('dv_augment_ingest', dtype, fr, fr.shape[0], fr.shape[1], fr.shape[2], tb, N, T, H, W, a, 4, pad,
             torch.tensor(mean).to(gpu), (1 / torch.tensor(std)).to(gpu), None if perm is None else perm.to(gpu),
             0 if perm is None else perm.shape[1], scratch, None if blur is None else bl, None if blur is None else btmp)
    y = ops.act_to_ncdhw(a)                                                        # [N, 3, T, H + 2 pad, W + 2 pad]
    if pad:
        inner = y[:, :, :, pad:-pad, pad:-pad]
        assert float(y.abs().sum() - inner.abs().sum()) == 0.0                     # the border stays zero
        y = inner
    assert float(a.buf[:, 3].abs().max()) == 0.0
    return y


@pytest.mark.parametrize('dtype,pad', [(DV_F32, 0), (DV_F32, 3), (DV_BF16, 3)])
def test_augment_ingest_against_reference_fixture(gpu, dtype, pad):
    """dv_augment_ingest == the reference's utils/transforms.py functions on the fixture rows (every op, order, resize,
    flip; tests/golden/augment.npz): crop / hflip / resize / adjust_* / rgb_to_grayscale / normalize"""
    from tests.util import gold
    from dualvar_amd.utils.transforms import AUG_ROW
    g = gold('augment')
    H, W = (int(v) for v in g['HW'])
    table = np.ascontiguousarray(g['A/table']).view(AUG_ROW).reshape(-1)
    want = torch.from_numpy(g['A/want']).permute(0, 2, 1, 3, 4)                    # [clips, 3, T, H, W]
    got = _augment(gpu, dtype, g['frames'], table, want.shape[0], want.shape[2], H, W, g['mean'].tolist(), g['std'].tolist(),
                   pad=pad).cpu()
    err = float((got - want).abs().max())
    print(f'augment ingest vs reference: max abs err {err:.2e} (values up to {float(want.abs().max()):.2f})')
    assert err < (2e-5 if dtype == DV_F32 else 2e-2)


@pytest.mark.parametrize('dtype,pad', [(DV_F32, 0), (DV_BF16, 3)])
def test_augment_gaussian_blur_against_pil_fixture(gpu, dtype, pad):
    """The SimCLR Gaussian blur of the ingest (utils/augmentation.py:706-721 -> PIL ImageFilter.GaussianBlur on the uint8
    frame): tests/golden/augment.npz part D holds PIL's own output (oracle/gen_golden.py ran Pillow 12.2.0 on the
    colour-jittered, re-quantised frames).  The integer kernel reproduces it BIT FOR BIT; frames without a blur row pass
    through the ordinary path."""
    from tests.util import gold
    from dualvar_amd.utils.transforms import AUG_BLUR, AUG_ROW, box_blur_params
    g = gold('augment')
    H, W = (int(v) for v in g['HW'])
    table = np.ascontiguousarray(g['D/table']).view(AUG_ROW).reshape(-1)
    blur = np.zeros(len(table), dtype=AUG_BLUR)
    for n, sg in enumerate(g['D/sigma']):
        if sg > 0:
            blur['radius'][n], blur['ww'][n], blur['fw'][n] = box_blur_params(float(sg))
    assert np.array_equal(blur.view(np.uint8).reshape(-1, 16), g['D/blur'])             # host parameters == the oracle's
    T_ = int(g['D/T'])
    N = len(table) // T_
    got = _augment(gpu, dtype, g['frames'], table, N, T_, H, W, g['mean'].tolist(), g['std'].tolist(), pad=pad, blur=blur).cpu()
    want_u8 = g['D/want_u8']                                                            # [N*T, H, W, 3] PIL's uint8 frames
    mean, std = torch.tensor(g['mean']).view(1, 3, 1, 1, 1), torch.tensor(g['std']).view(1, 3, 1, 1, 1)
    on = torch.from_numpy(g['D/sigma'] > 0).view(N, 1, T_, 1, 1)
    assert bool(on.any()) and not bool(on.all())
    if dtype == DV_F32:
        back = torch.round((got * std + mean) * 255.0).to(torch.int64)                  # undo Normalize and ToTensor
        want = torch.from_numpy(want_u8).view(N, T_, H, W, 3).permute(0, 4, 1, 2, 3).to(torch.int64)
        d = (back - want).abs() * on
        print('blurred frames vs PIL: max |difference| in uint8 steps', int(d.max()), '; pixels off by one step:', int((d > 0).sum()),
              'of', int(on.sum()) * 3 * H * W)
        # clip 0 is a plain crop: its quantised frame is the source bytes, so the integer blur must match PIL bit for bit.
        # Behind colour ops / a bilinear resize the float frame can sit within one ulp of a byte boundary, where the GPU's
        # and the CPU's mul(255).byte() truncate differently: such an input byte moves the blurred output by at most one step.
        assert float(g['D/sigma'][0]) > 0 and int(d[0].max()) == 0
        assert int(d.max()) <= 1 and int((d > 0).sum()) <= 0.01 * int(on.sum()) * 3 * H * W
    want_f = (torch.from_numpy(g['D/want']).view(N, T_, 3, H, W).permute(0, 2, 1, 3, 4) - mean) / std
    e = (got - want_f).abs()
    step = float((1 / 255.0 / std).max())                                             # one uint8 step after Normalize
    tol = 2e-5 if dtype == DV_F32 else 2e-2
    # every element within tolerance, except the few truncation-boundary pixels of blurred frames (one step off, see above)
    off = (e > tol)
    assert float(e.max()) <= step + tol and int(off.sum()) <= 0.01 * e.numel() and not bool((off & ~on.expand_as(off)).any())


def test_augment_hue_against_reference_fixture(gpu):
    """DV_AUG_HUE on the GPU == the reference's adjust_hue_np (uint8 result in tests/golden/augment.npz, part C)"""
    from tests.util import gold
    from tests.test_oracle_golden import _hue_matches
    from dualvar_amd.utils.transforms import AUG_ROW
    g = gold('augment')
    H, W = (int(v) for v in g['HW'])
    t = np.ascontiguousarray(g['C/table']).view(AUG_ROW).reshape(-1)
    a = ops.new_act(len(t), 1, H, W, 3, DV_F32, gpu, cpitch=4, zero=True)
    fr = torch.from_numpy(np.ascontiguousarray(g['frames'])).to(gpu)
    tb = torch.from_numpy(t.view(np.uint8).copy()).to(gpu)
    ops.call('dv_augment_ingest', DV_F32, fr, fr.shape[0], fr.shape[1], fr.shape[2], tb, len(t), 1, H, W, a, 4, 0, None, None, None, 0,
             torch.empty(len(t), device=gpu), None, None)
    got = ops.act_to_ncdhw(a)[:, :, 0].permute(0, 2, 3, 1).cpu().numpy()
    bad, boundary = _hue_matches(got.astype(np.float64) * 255.0, g['C/want_u8'])
    print(f'hue vs adjust_hue_np: {bad} mismatching pixels, {boundary} truncation-boundary cases of {got.size}')
    assert bad == 0 and boundary <= 10


def test_augment_ingest_full_size_rows_against_oracle(gpu):
    """128x171 decoded frames -> 112x112 windows (plain and resized crops, flips, all colour-op orders, segment
    shuffle) at the bench's frame size, against oracle/augment_ref.py"""
    from oracle import augment_ref as A
    r = np.random.RandomState(11)
    n_src, Hs, Ws, N, T, H, W = 10, 128, 171, 3, 8, 112, 112
    yy, xx = np.mgrid[0:Hs, 0:Ws]
    frames = (((np.sin(yy / 9.0)[..., None] * np.cos(xx / 7.0)[..., None] * 0.4 + 0.5)[None] * r.uniform(0.4, 1, (n_src, 1, 1, 3))
               + r.uniform(-0.2, 0.2, (n_src, Hs, Ws, 3))).clip(0, 1) * 255).round().astype(np.uint8)
    table = np.zeros(N * T, dtype=A.ROW)
    for f in range(N * T):
        row = table[f]
        row['src'] = r.randint(n_src)
        if f % 3 == 0:
            row['crop_h'], row['crop_w'] = H, W
        else:
            row['crop_h'], row['crop_w'] = r.randint(60, Hs + 1), r.randint(80, Ws + 1)
        row['crop_i'], row['crop_j'] = r.randint(0, Hs - row['crop_h'] + 1), r.randint(0, Ws - row['crop_w'] + 1)
        row['flip'] = r.randint(2)
        codes = list(r.permutation([A.BRIGHTNESS, A.CONTRAST, A.SATURATION, A.GRAY, A.HUE]))[:r.randint(0, 6)]
        for k, c in enumerate(codes):
            row['op'][k], row['factor'][k] = c, 1.0 if c == A.GRAY else r.uniform(-0.3, 0.3) if c == A.HUE else r.uniform(0.2, 1.8)
    perm = torch.tensor([[1, 0], [0, 1], [1, 0]], dtype=torch.int32)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    got = _augment(gpu, DV_F32, frames, table, N, T, H, W, mean, std, perm=perm, pad=3).cpu()
    want = A.augment_ingest(frames, table, N, T, H, W, mean, std, perm=perm.numpy())
    err = float((got - want).abs().max())
    print(f'augment ingest 112x112 vs oracle: max abs err {err:.2e}')
    assert err < 3e-5
    # an out-of-range row is clamped into the source instead of faulting
    bad = table.copy()
    bad['src'][0], bad['crop_i'][1], bad['crop_h'][2] = 10 ** 6, -5, 10 ** 6
    assert torch.isfinite(_augment(gpu, DV_F32, frames, bad, N, T, H, W, mean, std)).all()


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
@pytest.mark.parametrize('kt,st,pt', [(1, 1, 0), (3, 1, 1)])
def test_rgb_stem_as_pixel_pair_conv(gpu, dtype, kt, st, pt):
    """dv_ingest_ncdhw_pad + the (kt,7,4)-tap stride-(st,2,1) conv over 8-channel pixel pairs == the reference's
    7x7 / stride 2 / padding 3 RGB stem (s3dg.py:137, r21d.py:201, r3d.py:116), forward and weight gradient; the
    structural zero tap (kw = 7 of the 8-wide rows) gets a gradient from the gather and dv_fill_cols_f32 clears it."""
    N, T, H, W, O = 2, 4, 18, 22, 24
    x = q(rnd(N, 3, T, H, W, seed=61), dtype)
    w = q(0.2 * rnd(O, 3, kt, 7, 7, seed=62), dtype)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, (st, 2, 2), (pt, 3, 3))
    gy = q(rnd(*yr.shape, seed=63), dtype)
    yr.backward(gy)

    a = ops.new_act(N, T, H + 6, W + 6, 3, dtype, gpu, cpitch=4, zero=True)
    ops.call('dv_ingest_ncdhw_pad', dtype, x.to(gpu), a, N, 3, T, H, W, 3 * T * H * W, 4, None, None, None, 0, 3)
    frames = a.buf.view(N, T, H + 6, W + 6, 4).float().cpu()
    assert torch.equal(frames[:, :, 3:-3, 3:-3, :3].permute(0, 4, 1, 2, 3), x)
    assert float(frames[:, :, :3].abs().max()) == 0 and float(frames[:, :, -3:].abs().max()) == 0
    assert float(frames[:, :, :, :3].abs().max()) == 0 and float(frames[:, :, :, -3:].abs().max()) == 0

    pairs = ops.Act(a.buf.view(-1, 8), N, T, H + 6, (W + 6) // 2, 8, 8, 0, dtype, 8)
    To, Ho, Wo = yr.shape[2:]
    y = ops.new_act(N, To, Ho, Wo, O, dtype, gpu)
    d = ops.conv_desc(dtype, pairs, y, (kt, 7, 4), (st, 2, 1), (pt, 0, 0))
    w8 = torch.zeros(O, 3, kt, 7, 8)
    w8[..., :7] = w
    wp = ops.pack_weight(w8, 4).to(gpu)                       # [O][kt*7*8 taps][4] == [O][kt*7*4 pair taps][8]
    wq = wp if dtype == DV_F32 else wp.bfloat16()
    ops.conv_fwd(d, pairs, wq, None, y, None)
    close(ops.act_to_ncdhw(y), yr.detach(), dtype, 'stem fwd')

    dy = ops.act_from_ncdhw(gy.to(gpu), dtype)
    dw = torch.zeros(O, kt * 7 * 8, 4, device=gpu)
    ops.conv_wgrad(d, pairs, dy, dw)
    assert float(dw.view(O, 3 if False else kt, 7, 8, 4)[:, :, :, 7].abs().max()) > 0      # the pad tap saw real pixels
    ops.call('dv_fill_cols_f32', dw, O * kt * 7, 32, 28, 4, 0.0)
    got = dw.view(O, kt, 7, 8, 4).cpu()
    assert float(got[:, :, :, 7].abs().max()) == 0 and float(got[..., 3].abs().max()) == 0
    close(got[:, :, :, :7, :3].permute(0, 4, 1, 2, 3), wr.grad, dtype, 'stem wgrad')


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
@pytest.mark.parametrize('C_,residual,relu', [(64, False, True), (83, True, True), (24, True, False), (1152, False, True)])
def test_batchnorm_fwd_bwd(gpu, dtype, C_, residual, relu):
    N, T, H, W = 3, 2, 5, 4
    x = q(rnd(N, C_, T, H, W, seed=5) * 2 + 0.5, dtype)
    res = q(rnd(N, C_, T, H, W, seed=6), dtype) if residual else None
    gamma, beta = 1 + 0.2 * rnd(C_, seed=7), 0.1 * rnd(C_, seed=8)
    bn = torch.nn.BatchNorm3d(C_)
    bn.weight.data.copy_(gamma)
    bn.bias.data.copy_(beta)
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if residual else None
    o = bn(xr)
    if residual:
        o = o + rr
    yr = F.relu(o) if relu else o
    gy = q(rnd(*yr.shape, seed=9), dtype)
    yr.backward(gy)

    M = N * T * H * W
    xa = ops.act_from_ncdhw(x.to(gpu), dtype)
    # statistics straight from the tensor (as the conv epilogue would emit them): one 128-row tile each
    tiles = (M + 127) // 128
    xs = xa.buf[:, :C_].float()
    part = torch.zeros(2, C_, tiles, device=gpu)               # [2][C][tiles], as dv_conv3d_fwd writes them
    for i in range(tiles):
        blk = xs[i * 128:(i + 1) * 128]
        part[0, :, i] = blk.sum(0)
        part[1, :, i] = ((blk - blk.mean(0)) ** 2).sum(0)
    local = torch.zeros(2 * C_ + 1, device=gpu)
    ops.call('dv_bn_reduce_stats', part, tiles, 128, C_, M, C_, local)
    # two "ranks" with half the data each must give the same result as one (SyncBN identity)
    rm, rv = torch.zeros(C_, device=gpu), torch.ones(C_, device=gpu)
    CP = ops.cp8(C_)

    def padded(t):                  # per-channel arrays are read with 16-byte loads up to round_up(C, 8)
        o = torch.zeros(CP, device=gpu)
        o[:C_] = t.to(gpu)
        return o
    mean, invstd, scale, shift = (torch.zeros(CP, device=gpu) for _ in range(4))
    gam, bet = padded(gamma), padded(beta)
    ops.call('dv_bn_finalize', local, 1, 2 * C_ + 1, C_, gam, bet, 1e-5, 0.1, rm, rv, mean, invstd, scale, shift)
    # the fused single-rank variant must agree with reduce + finalize
    rm2, rv2, local2 = torch.zeros(C_, device=gpu), torch.ones(C_, device=gpu), torch.zeros(2 * C_ + 1, device=gpu)
    o2 = [torch.zeros(CP, device=gpu) for _ in range(4)]
    ops.call('dv_bn_stats_finalize', part, tiles, 128, C_, M, C_, local2, gam, bet, 1e-5, 0.1, rm2, rv2, *o2)
    for a_, b_ in zip((mean, invstd, scale, shift, rm, rv, local), (*o2, rm2, rv2, local2)):
        assert torch.equal(a_, b_)
    close(rm, bn.running_mean, DV_F32, 'running_mean', factor=10)
    close(rv, bn.running_var, DV_F32, 'running_var', factor=10)
    ya = ops.new_act(N, T, H, W, C_, dtype, gpu)
    ra = ops.act_from_ncdhw(res.to(gpu), dtype) if residual else None
    ops.call('dv_bn_apply', dtype, xa, xa.ld, scale, shift, ra, ra.ld if ra else 0, ya, ya.ld, M, C_, ops.DV_RELU if relu else 0)
    close(ops.act_to_ncdhw(ya), yr, dtype, 'bn apply')

    dya = ops.act_from_ncdhw(gy.to(gpu), dtype)
    flags = 0 if relu else ops.DV_NO_RELU_MASK
    sums = torch.zeros(4, 2, CP, device=gpu)          # 4 replicas of the atomic accumulators
    ops.call('dv_bn_bwd_reduce', dtype, dya, dya.ld, ya, ya.ld, xa, xa.ld, mean, invstd, M, C_, flags, sums, 4, None)
    # ordered form: per-block partials + ticket, summed in block order by the last block -- the same sums, twice the same bits
    from dualvar_amd import _lib as L_
    wsn = int(L_.load().dv_bn_bwd_reduce_workspace(M, C_)) // 4
    ws = torch.zeros(wsn, device=gpu)
    so = [torch.full((2, CP), 7.0, device=gpu) for _ in range(2)]
    for t in so:
        ops.call('dv_bn_bwd_reduce', dtype, dya, dya.ld, ya, ya.ld, xa, xa.ld, mean, invstd, M, C_, flags, t, 1, ws)
    assert torch.equal(so[0], so[1]) and float(ws[-8:].abs().max()) == 0.0
    close(so[0][:, :C_], sums.sum(0)[:, :C_], DV_F32, 'ordered bn bwd sums', factor=20)
    dg, db = torch.zeros(C_, device=gpu), torch.zeros(C_, device=gpu)
    dxa = ops.new_act(N, T, H, W, C_, dtype, gpu)
    dra = ops.new_act(N, T, H, W, C_, dtype, gpu) if residual else None
    ops.call('dv_bn_bwd_apply', dtype, dya, dya.ld, ya, ya.ld, xa, xa.ld, mean, invstd, gam, sums, 4,
             1.0 / M, 1.0, dg, db, dxa, dxa.ld, dra, dra.ld if dra else 0, M, C_, flags)
    f = 3 if dtype == DV_BF16 else 20
    close(ops.act_to_ncdhw(dxa), xr.grad, dtype, 'bn dx', factor=f)
    close(dg, bn.weight.grad, dtype, 'dgamma', factor=f)
    close(db, bn.bias.grad, dtype, 'dbeta', factor=f)
    if residual:
        close(ops.act_to_ncdhw(dra), rr.grad, dtype, 'dres')


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
@pytest.mark.parametrize('k,s,p,shape', [((1, 3, 3), (1, 2, 2), (0, 1, 1), (3, 2, 14, 10)), ((3, 3, 3), (2, 2, 2), (1, 1, 1), (2, 5, 9, 8)),
                                        ((1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 3, 9, 7))])
def test_batchnorm_relu_maxpool_fused(gpu, dtype, k, s, p, shape):
    """dv_bn_apply_maxpool / dv_bn_bwd_*_maxpool == the two-pass form they replace (dv_bn_apply + dv_maxpool3d_fwd;
    dv_maxpool3d_bwd + dv_bn_bwd_reduce / apply with the mask from x): forward bit for bit, backward up to the order of
    the fp32 atomics"""
    from dualvar_amd._lib import DV_MASK_FROM_X
    N, T, H, W = shape
    C_ = 72
    x = q(rnd(N, C_, T, H, W, seed=31) * 2 + 0.3, dtype)
    M = N * T * H * W
    CP = ops.cp8(C_)
    xa = ops.act_from_ncdhw(x.to(gpu), dtype)
    xs = xa.buf[:, :C_].float()
    mean_t, var_t = xs.mean(0), xs.var(0, unbiased=False)
    gamma, beta = 1 + 0.2 * rnd(C_, seed=32).to(gpu), 0.1 * rnd(C_, seed=33).to(gpu)

    def padded(t):
        o = torch.zeros(CP, device=gpu)
        o[:C_] = t
        return o
    invstd_t = torch.rsqrt(var_t + 1e-5)
    mean, invstd, gam = padded(mean_t), padded(invstd_t), padded(gamma)
    scale, shift = padded(gamma * invstd_t), padded(beta - mean_t * gamma * invstd_t)
    # --- two passes
    ya = ops.new_act(N, T, H, W, C_, dtype, gpu)
    ops.call('dv_bn_apply', dtype, xa, xa.ld, scale, shift, None, 0, ya, ya.ld, M, C_, ops.DV_RELU)
    To, Ho, Wo = ops.conv_out_dims(ya, k, s, p)
    pa = ops.new_act(N, To, Ho, Wo, C_, dtype, gpu)
    d = ops.pool_desc(dtype, ya, pa, k, s, p)
    idx = torch.zeros(pa.rows, CP, dtype=torch.uint8, device=gpu)
    ops.call('dv_maxpool3d_fwd', d, ya, pa, idx)
    # --- fused
    pb = ops.new_act(N, To, Ho, Wo, C_, dtype, gpu)
    idx2 = torch.zeros_like(idx)
    d2 = ops.pool_desc(dtype, xa, pb, k, s, p)
    ops.call('dv_bn_apply_maxpool', d2, xa, scale, shift, pb, idx2)
    assert torch.equal(pa.buf, pb.buf) and torch.equal(idx[:, :C_], idx2[:, :C_])
    # --- backward
    gp = ops.act_from_ncdhw(q(rnd(N, C_, To, Ho, Wo, seed=34), dtype).to(gpu), dtype)
    dya = ops.new_act(N, T, H, W, C_, dtype, gpu)
    ops.call('dv_maxpool3d_bwd', d, gp, idx, dya, 0)
    sums = torch.zeros(4, 2, CP, device=gpu)
    ops.call('dv_bn_bwd_reduce', dtype, dya, dya.ld, ya, ya.ld, xa, xa.ld, mean, invstd, M, C_, 0, sums, 4, None)
    dg, db = torch.zeros(C_, device=gpu), torch.zeros(C_, device=gpu)
    dxa = ops.new_act(N, T, H, W, C_, dtype, gpu)
    ops.call('dv_bn_bwd_apply', dtype, dya, dya.ld, ya, ya.ld, xa, xa.ld, mean, invstd, gam, sums, 4, 1.0 / M, 1.0, dg, db,
             dxa, dxa.ld, None, 0, M, C_, 0)
    sums2 = torch.zeros(4, 2, CP, device=gpu)
    ops.call('dv_bn_bwd_reduce_maxpool', d2, gp, idx2, xa, mean, invstd, scale, shift, sums2, 4)
    dg2, db2 = torch.zeros(C_, device=gpu), torch.zeros(C_, device=gpu)
    dxb = ops.new_act(N, T, H, W, C_, dtype, gpu)
    ops.call('dv_bn_bwd_apply_maxpool', d2, gp, idx2, xa, mean, invstd, gam, scale, shift, sums2, 4, 1.0 / M, 1.0, dg2, db2,
             dxb, dxb.ld)
    # (bf16: the two-pass form rounds dL/dy to bf16 between the pool and the BatchNorm backward, the fused one does not)
    close(sums2.sum(0), sums.sum(0), dtype, 'fused bn+pool sums', factor=1 if dtype == DV_BF16 else 5)
    close(dxb.buf.float(), dxa.buf.float(), dtype, 'fused bn+pool dx', factor=1 if dtype == DV_BF16 else 5)
    close(dg2, dg, dtype, 'fused bn+pool dgamma', factor=1 if dtype == DV_BF16 else 5)
    close(db2, db, dtype, 'fused bn+pool dbeta', factor=1 if dtype == DV_BF16 else 5)


def test_bn_two_rank_combine(gpu):
    C_, M = 40, 600
    x = rnd(M, C_, seed=11) * 3 + 1
    stats = []
    for part in (x[:250], x[250:]):
        stats.append(torch.cat([part.sum(0), ((part - part.mean(0)) ** 2).sum(0), torch.tensor([float(part.shape[0])])]))
    st = torch.stack(stats).to(gpu)
    outs = [torch.empty(C_, device=gpu) for _ in range(4)]
    ops.call('dv_bn_finalize', st, 2, 2 * C_ + 1, C_, torch.ones(C_, device=gpu), torch.zeros(C_, device=gpu), 1e-5, 0.1, None, None, *outs)
    close(outs[0], x.mean(0), DV_F32, 'mean 2-rank', factor=5)
    close(outs[1], (x.var(0, unbiased=False) + 1e-5).rsqrt(), DV_F32, 'invstd 2-rank', factor=5)
    # the host-side restatement used by the gloo tests agrees
    from dualvar_amd.parallel import combine_bn_stats
    m_h, v_h = combine_bn_stats(st.cpu())
    close(outs[0], m_h, DV_F32, 'host combine mean', factor=5)


POOL_CASES = [((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((3, 3, 3), (1, 1, 1), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
              ((2, 2, 2), (2, 2, 2), (0, 0, 0))]


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
@pytest.mark.parametrize('k,s,p', POOL_CASES)
def test_maxpool(gpu, dtype, k, s, p):
    N, C_, T, H, W = 2, 24, 4, 9, 10
    x = F.relu(q(rnd(N, C_, T, H, W, seed=12), dtype))       # post-ReLU: many exact ties at 0
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, k, s, p)
    gy = q(rnd(*yr.shape, seed=13), dtype)
    yr.backward(gy)
    xa = ops.act_from_ncdhw(x.to(gpu), dtype)
    ya = ops.new_act(N, *yr.shape[2:], C_, dtype, gpu)
    idx = torch.zeros(ya.rows, ops.cp8(C_), dtype=torch.uint8, device=gpu)
    d = ops.pool_desc(dtype, xa, ya, k, s, p)
    ops.call('dv_maxpool3d_fwd', d, xa, ya, idx)
    close(ops.act_to_ncdhw(ya), yr, DV_F32, 'maxpool fwd')
    dya = ops.act_from_ncdhw(gy.to(gpu), dtype)
    dxa = ops.new_act(N, T, H, W, C_, dtype, gpu)
    ops.call('dv_maxpool3d_bwd', d, dya, idx, dxa, 0)
    close(ops.act_to_ncdhw(dxa), q(xr.grad, dtype), dtype, 'maxpool bwd')
    ops.call('dv_maxpool3d_bwd', d, dya, idx, dxa, ops.DV_ACCUM)
    close(ops.act_to_ncdhw(dxa), 2 * xr.grad, dtype, 'maxpool bwd accum', factor=2)


@pytest.mark.parametrize('W', [3, 7, 14])
def test_maxpool_333_fast_path_signed_values(gpu, W):
    """3x3x3 / stride 1 / padding 1 in bf16 orders values through integer keys (staged kernel; W = 3 takes the gather
    kernel): negative values and the uint8 tap index have to agree with PyTorch"""
    N, C_, T, H = 2, 40, 3, 5
    x = q(rnd(N, C_, T, H, W, seed=71) * 3 - 1, DV_BF16)
    x[:, :8] = -x[:, :8].abs()                                      # windows whose maximum is negative
    xr = x.clone().requires_grad_(True)
    yr, ir = F.max_pool3d(xr, 3, 1, 1, return_indices=True)
    gy = q(rnd(*yr.shape, seed=72), DV_BF16)
    yr.backward(gy)
    xa = ops.act_from_ncdhw(x.to(gpu), DV_BF16)
    ya = ops.new_act(N, T, H, W, C_, DV_BF16, gpu)
    idx = torch.zeros(ya.rows, ops.cp8(C_), dtype=torch.uint8, device=gpu)
    d = ops.pool_desc(DV_BF16, xa, ya, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    ops.call('dv_maxpool3d_fwd', d, xa, ya, idx)
    assert torch.equal(ops.act_to_ncdhw(ya).cpu(), yr.detach())
    # tap index -> flat input index, as PyTorch reports it
    tap = idx.view(N, T, H, W, -1)[..., :C_].permute(0, 4, 1, 2, 3).long().cpu()
    dt, dh, dw = tap // 9, tap // 3 % 3, tap % 3
    tt = torch.arange(T).view(1, 1, T, 1, 1) - 1 + dt
    hh = torch.arange(H).view(1, 1, 1, H, 1) - 1 + dh
    ww = torch.arange(W).view(1, 1, 1, 1, W) - 1 + dw
    assert torch.equal((tt * H + hh) * W + ww, ir)
    dya = ops.act_from_ncdhw(gy.to(gpu), DV_BF16)
    dxa = ops.new_act(N, T, H, W, C_, DV_BF16, gpu)
    ops.call('dv_maxpool3d_bwd', d, dya, idx, dxa, 0)
    close(ops.act_to_ncdhw(dxa), q(xr.grad, DV_BF16), DV_BF16, 'maxpool333 bwd')


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
@pytest.mark.parametrize('shape', [(3, 40, 4, 14, 14), (2, 72, 2, 7, 7), (2, 24, 1, 16, 30), (1, 36, 5, 9, 8), (2, 8, 3, 5, 5)])
def test_maxpool_333_staged_tiles(gpu, dtype, shape):
    """the LDS-staged 3x3x3 / stride 1 / padding 1 kernels (planes of 25 pixels and more): values, PyTorch's first-maximum
    tap index and the gathered gradient, on whole tiles, ragged tiles, several tiles per plane, partial channel chunks,
    T = 1 and T > 3 (ring wrap); the gradient sums run in the gather kernel's tap order, so DV_ACCUM on zeros is
    bit-identical to the plain call"""
    N, C_, T, H, W = shape
    x = q(rnd(N, C_, T, H, W, seed=81) * 3 - 1, dtype)
    x[:, :4] = -x[:, :4].abs()
    x[:, 4:8] = F.relu(x[:, 4:8])                                   # exact ties at 0
    xr = x.clone().requires_grad_(True)
    yr, ir = F.max_pool3d(xr, 3, 1, 1, return_indices=True)
    gy = q(rnd(*yr.shape, seed=82), dtype)
    yr.backward(gy)
    xa = ops.act_from_ncdhw(x.to(gpu), dtype)
    ya = ops.new_act(N, T, H, W, C_, dtype, gpu)
    idx = torch.full((ya.rows, ops.cp8(C_)), 77, dtype=torch.uint8, device=gpu)
    d = ops.pool_desc(dtype, xa, ya, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    ops.call('dv_maxpool3d_fwd', d, xa, ya, idx)
    assert torch.equal(ops.act_to_ncdhw(ya).cpu(), yr.detach())
    tap = idx.view(N, T, H, W, -1)[..., :C_].permute(0, 4, 1, 2, 3).long().cpu()
    dt, dh, dw = tap // 9, tap // 3 % 3, tap % 3
    tt = torch.arange(T).view(1, 1, T, 1, 1) - 1 + dt
    hh = torch.arange(H).view(1, 1, 1, H, 1) - 1 + dh
    ww = torch.arange(W).view(1, 1, 1, 1, W) - 1 + dw
    assert torch.equal((tt * H + hh) * W + ww, ir)
    dya = ops.act_from_ncdhw(gy.to(gpu), dtype)
    dxa = ops.new_act(N, T, H, W, C_, dtype, gpu)
    dxa.buf.fill_(float('nan'))
    ops.call('dv_maxpool3d_bwd', d, dya, idx, dxa, 0)
    close(ops.act_to_ncdhw(dxa), q(xr.grad, dtype), dtype, 'staged maxpool bwd')
    dxb = ops.new_act(N, T, H, W, C_, dtype, gpu, zero=True)
    ops.call('dv_maxpool3d_bwd', d, dya, idx, dxb, ops.DV_ACCUM)
    assert torch.equal(dxa.buf[:, :C_], dxb.buf[:, :C_])


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
def test_self_gating_and_mean(gpu, dtype):
    N, C_, T, H, W = 3, 48, 2, 5, 5
    S = T * H * W
    x = q(F.relu(rnd(N, C_, T, H, W, seed=14)), dtype)
    fc = torch.nn.Linear(C_, C_)
    xr = x.clone().requires_grad_(True)
    g_ref = torch.sigmoid(fc(xr.mean(dim=[2, 3, 4])))
    yr = g_ref[:, :, None, None, None] * xr
    gy = q(rnd(*yr.shape, seed=15), dtype)
    yr.backward(gy)

    xa = ops.act_from_ncdhw(x.to(gpu), dtype)
    mean = torch.empty(N, C_, device=gpu)
    ops.call('dv_spatial_mean', dtype, xa, xa.ld, N, S, C_, mean)
    close(mean, x.mean(dim=[2, 3, 4]), DV_F32, 'spatial mean', factor=5)
    # fc + sigmoid through the strided fp32 GEMM
    Wt, b = fc.weight.detach().to(gpu), fc.bias.detach().to(gpu)
    pre = b.repeat(N, 1).contiguous()
    ops.call('dv_gemm_f32', N, C_, C_, mean, C_, 1, Wt, 1, C_, pre, C_, 1.0, 1)
    g = torch.sigmoid(pre)                      # test-side only; the product fuses sigmoid in the conv epilogue
    close(g, g_ref, DV_F32, 'gate', factor=10)
    ya = ops.new_act(N, T, H, W, C_, dtype, gpu)
    ops.call('dv_gate_scale', dtype, xa, xa.ld, g, N, S, C_, ya, ya.ld)
    close(ops.act_to_ncdhw(ya), yr, dtype, 'gate scale')
    dya = ops.act_from_ncdhw(gy.to(gpu), dtype)
    dpre = torch.empty(N, C_, device=gpu)
    ops.call('dv_gate_bwd_reduce', dtype, dya, dya.ld, xa, xa.ld, g, N, S, C_, dpre, 0)
    # in-place gating variant: the second operand is the gated output
    dpre2 = torch.empty(N, C_, device=gpu)
    ops.call('dv_gate_bwd_reduce', dtype, dya, dya.ld, ya, ya.ld, g, N, S, C_, dpre2, 1)
    close(dpre2, dpre, dtype, 'gate dpre from output', factor=2)
    # grouped GEMM == the two plain GEMMs
    import ctypes as C
    from dualvar_amd import _lib as L
    arr = (L.GemmDesc * 2)()
    outA, outB = torch.zeros(N, C_, device=gpu), torch.zeros(C_, C_, device=gpu)
    for i, dd in enumerate([dict(A=dpre.data_ptr(), sam=C_, sak=1, B=Wt.data_ptr(), sbk=C_, sbn=1, C=outA.data_ptr(), ldc=C_, bias=0, M=N, N=C_, K=C_, flags=0, alpha=1.0),
                            dict(A=dpre.data_ptr(), sam=1, sak=C_, B=mean.data_ptr(), sbk=C_, sbn=1, C=outB.data_ptr(), ldc=C_, bias=0, M=C_, N=C_, K=N, flags=ops.DV_ACCUM, alpha=1.0)]):
        for k, v in dd.items():
            setattr(arr[i], k, v)
    t0 = ((N + 31) // 32) * ((C_ + 31) // 32)
    arr[0].tile_end, arr[1].tile_end = t0, t0 + ((C_ + 31) // 32) ** 2
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(gpu)
    ops.call('dv_gemm_f32_grouped', tab, 2, int(arr[1].tile_end))
    dmean = torch.empty(N, C_, device=gpu)
    ops.call('dv_gemm_f32', N, C_, C_, dpre, C_, 1, Wt, C_, 1, dmean, C_, 1.0, 0)
    dxa = ops.new_act(N, T, H, W, C_, dtype, gpu)
    ops.call('dv_gate_bwd_apply', dtype, dya, dya.ld, g, dmean, N, S, C_, dxa, dxa.ld, 0)
    close(ops.act_to_ncdhw(dxa), xr.grad, dtype, 'gate dx', factor=2)
    dW = torch.zeros(C_, C_, device=gpu)
    ops.call('dv_gemm_f32', C_, C_, N, dpre, 1, C_, mean, C_, 1, dW, C_, 1.0, 1)
    close(dW, fc.weight.grad, dtype, 'gate dW', factor=2)
    close(outA, dmean, DV_F32, 'grouped gemm 0')
    close(outB, dW, DV_F32, 'grouped gemm 1')
    dbias = torch.zeros(C_, device=gpu)
    ops.call('dv_colsum_f32', dpre, C_, N, C_, dbias)
    close(dbias, fc.bias.grad, dtype, 'gate db', factor=2)
    # spatial mean backward
    dm = rnd(N, C_, seed=16).to(gpu)
    ops.call('dv_spatial_mean_bwd', dtype, dm, N, S, C_, dxa, dxa.ld, 0)
    ref = (dm.cpu() / S)[:, :, None, None, None].expand(N, C_, T, H, W)
    close(ops.act_to_ncdhw(dxa), q(ref, dtype), dtype, 'mean bwd')


def test_l2norm_relu_colsum(gpu):
    x = rnd(10, 128, seed=17)
    xr = x.clone().requires_grad_(True)
    yr = F.normalize(xr, dim=1)
    gy = rnd(10, 128, seed=18)
    yr.backward(gy)
    xg = x.to(gpu)
    y, nrm, dx = torch.empty_like(xg), torch.empty(10, device=gpu), torch.empty_like(xg)
    ops.call('dv_l2norm_fwd', xg, 10, 128, 1e-12, y, nrm)
    ops.call('dv_l2norm_bwd', gy.to(gpu), y, nrm, 10, 128, dx)
    close(y, yr, DV_F32, 'l2norm')
    close(dx, xr.grad, DV_F32, 'l2norm bwd', factor=5)
    r = torch.empty_like(xg)
    ops.call('dv_relu_bwd_f32', gy.to(gpu), xg, x.numel(), r)
    close(r, gy * (x > 0), DV_F32, 'relu bwd')


def test_losses_against_oracle(gpu):
    from oracle import torch_ref as O
    import types
    N, dim = 6, 128
    m = O.SimCLR_TimeSeriesV4.__new__(O.SimCLR_TimeSeriesV4)
    torch.nn.Module.__init__(m)
    m.distributed, m.T, m.aligned_T, m.n_series, m.series_dim, m.dim = False, 0.07, 0.07, 2, 64, 128
    m.args = types.SimpleNamespace(shufflerank_theta=0.05)
    feats = F.normalize(rnd(N, 2, dim, seed=19), dim=-1).requires_grad_(True)
    r = m.calc_clip_contrast_loss(feats, 2)
    r['clip_contrast_loss'].backward()
    fvm = feats.detach().permute(1, 0, 2).reshape(2 * N, dim).contiguous().to(gpu)     # view-major
    logits = torch.empty(2 * N, 2 * N - 1, device=gpu)
    loss_rows, rank0 = torch.empty(2 * N, device=gpu), torch.empty(2 * N, dtype=torch.int32, device=gpu)
    dsim = torch.empty(2 * N, 2 * N, device=gpu)
    ops.call('dv_ntxent_fwd', fvm, fvm, 2 * N, N, N, dim, 0, 1 / 0.07, logits, loss_rows, rank0, dsim)
    close(logits, r['clip_logits'], DV_F32, 'clip logits', factor=5)
    close(loss_rows.mean(), r['clip_contrast_loss'], DV_F32, 'clip loss', factor=5)
    top1 = O.calc_topk_accuracy(r['clip_logits'], r['clip_labels'], (1,))[0]
    assert abs(float((rank0 < 1).float().mean()) - float(top1)) < 1e-6
    # dF = dsim.F + dsim^T.F
    dF = torch.zeros(2 * N, dim, device=gpu)
    ops.call('dv_gemm_f32', 2 * N, dim, 2 * N, dsim, 2 * N, 1, fvm, dim, 1, dF, dim, 1.0, 0)
    ops.call('dv_gemm_f32', 2 * N, dim, 2 * N, dsim, 1, 2 * N, fvm, dim, 1, dF, dim, 1.0, 1)
    ref_g = feats.grad.permute(1, 0, 2).reshape(2 * N, dim)
    close(dF, ref_g, DV_F32, 'clip dF', factor=20)

    # tc head: series-mean vectors
    ser = F.normalize(rnd(N, 2, 2, 64, seed=20), dim=-1).requires_grad_(True)
    r2 = m.calc_tc_contrast_loss(ser)
    r2['tc_contrast_loss'].backward()
    svm = ser.detach().permute(1, 0, 2, 3).reshape(2 * N, 2, 64).contiguous().to(gpu)
    sm = torch.empty(2 * N, 64, device=gpu)
    ops.call('dv_group_mean_f32', svm, 2 * N, 2, 64, sm)
    ops.call('dv_ntxent_fwd', sm, sm, 2 * N, N, N, 64, 0, 1 / 0.07, logits, loss_rows, rank0, dsim)
    close(logits, r2['tc_logits'], DV_F32, 'tc logits', factor=5)
    close(loss_rows.mean(), r2['tc_contrast_loss'], DV_F32, 'tc loss', factor=5)
    dM = torch.zeros(2 * N, 64, device=gpu)
    ops.call('dv_gemm_f32', 2 * N, 64, 2 * N, dsim, 2 * N, 1, sm, 64, 1, dM, 64, 1.0, 0)
    ops.call('dv_gemm_f32', 2 * N, 64, 2 * N, dsim, 1, 2 * N, sm, 64, 1, dM, 64, 1.0, 1)
    dS = torch.empty(2 * N, 2, 64, device=gpu)
    ops.call('dv_group_mean_bwd_f32', dM, 2 * N, 2, 64, dS)
    close(dS, ser.grad.permute(1, 0, 2, 3).reshape(2 * N, 2, 64), DV_F32, 'tc dF', factor=20)

    # shuffle-rank margin (SimCLR: theta .05, clip 5; MoCo: no clip)
    for clip in (5.0, 0.0):
        rk = F.normalize(rnd(N, 2, 2, 64, seed=21), dim=-1).requires_grad_(True)     # [Bn, s, 2, D]
        r3 = O.ranking_loss(rk, 2, 2, 'x_', 0.5, theta=0.05, clip=clip if clip > 0 else None)
        r3['x_margin_contrast_loss'].backward()
        fv = rk.detach().permute(0, 2, 1, 3).reshape(N, 4, 64).contiguous().to(gpu)
        lg, ls, df, scr = torch.empty(N * 4, 3, device=gpu), torch.empty(1, device=gpu), torch.empty(N, 4, 64, device=gpu), torch.empty(N, device=gpu)
        ops.call('dv_rank_margin', fv, N, 2, 64, 0.05, clip, 0.5, lg, ls, df, scr)
        close(lg, r3['x_margin_logits'], DV_F32, 'rank logits', factor=5)
        close(ls[0], r3['x_margin_contrast_loss'], DV_F32, 'rank loss', factor=20)
        close(df.reshape(N, 2, 2, 64).permute(0, 2, 1, 3), rk.grad, DV_F32, 'rank grad', factor=50)


@pytest.mark.parametrize('B,K', [(5, 96), (32, 65536), (40, 4096)])
def test_infonce_against_oracle(gpu, B, K):
    """K = 65 536 is MoCo's queue (moco.py:79-81): dq = dlogits . queue^T then takes the split-K path of dv_gemm_f32"""
    D = 128
    qf = F.normalize(rnd(B, D, seed=22), dim=1).requires_grad_(True)
    kf = F.normalize(rnd(B, D, seed=23), dim=1)
    queue = F.normalize(rnd(D, K, seed=24), dim=0)
    pos = torch.einsum('nc,nc->n', [qf, kf]).unsqueeze(-1)
    neg = torch.einsum('nc,ck->nk', [qf, queue])
    lr = torch.cat([pos, neg], 1) / 0.07
    loss = F.cross_entropy(lr, torch.zeros(B, dtype=torch.long))
    loss.backward()
    logits, dl = torch.empty(B, K + 1, device=gpu), torch.empty(B, K + 1, device=gpu)
    lrows, rk, dq = torch.empty(B, device=gpu), torch.empty(B, dtype=torch.int32, device=gpu), torch.empty(B, D, device=gpu)
    ops.call('dv_infonce_fwd', qf.detach().to(gpu), kf.to(gpu), queue.to(gpu), B, D, K, 1 / 0.07, logits, lrows, rk, dl, dq, None, 0)
    # with the workspace the K-split of dq is ordered (partial tiles + tickets): same value, identical bits run to run
    from dualvar_amd import _lib as L_
    wsb = int(L_.load().dv_infonce_workspace(B, D, K))
    assert (wsb > 0) == (K >= 4096)
    if wsb:
        ws = torch.zeros(wsb // 4, device=gpu)
        dqo = [torch.full((B, D), 3.0, device=gpu) for _ in range(2)]
        for t in dqo:
            ops.call('dv_infonce_fwd', qf.detach().to(gpu), kf.to(gpu), queue.to(gpu), B, D, K, 1 / 0.07, logits, lrows, rk, dl, t, ws, wsb)
        assert torch.equal(dqo[0], dqo[1])
        close(dqo[0], qf.grad, DV_F32, 'infonce dq (ordered split-K)', factor=20)
    close(logits, lr, DV_F32, 'infonce logits', factor=5)
    close(lrows.mean(), loss, DV_F32, 'infonce loss', factor=5)
    close(dq, qf.grad, DV_F32, 'infonce dq', factor=20)


@pytest.mark.parametrize('M,N,K,relu', [(5, 40, 100, True), (128, 1024, 1024, True), (33, 65, 7, False), (128, 128, 1024, False)])
def test_gemm_ex_bias_relu(gpu, M, N, K, relu):
    """dv_gemm_f32_ex (descriptor by value: the heads' Linear / 1x1x1 conv forward) == x @ W^T + b (+ReLU) in fp64, with W
    stored at a wider pitch and C written into a wider buffer"""
    import ctypes as C
    from dualvar_amd import _lib as L
    x, w, b = rnd(M, K, seed=41), rnd(N, K + 8, seed=42), rnd(N, seed=43)
    want = x.double() @ w[:, :K].double().t() + b.double()
    if relu:
        want = want.clamp_min(0)
    xg, wg, bg = x.to(gpu), w.to(gpu), b.to(gpu)
    y = torch.full((M, N + 3), 7.0, device=gpu)
    d = L.GemmDesc()
    d.A, d.B, d.C, d.bias = xg.data_ptr(), wg.data_ptr(), y.data_ptr(), bg.data_ptr()
    d.sam, d.sak, d.sbk, d.sbn, d.ldc = K, 1, 1, K + 8, N + 3
    d.M, d.N, d.K, d.flags, d.alpha = M, N, K, (ops.DV_RELU if relu else 0), 1.0
    L.check(L.load().dv_gemm_f32_ex(C.byref(d), ops.stream_ptr()), 'dv_gemm_f32_ex')
    got = y.cpu()
    assert float((got[:, :N].double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6
    assert float((got[:, N:] - 7.0).abs().max()) == 0.0                        # nothing written beyond N
    # the plain entry point takes the same kernel for products with few tiles: A^T B form, accumulate
    acc = torch.ones(K, N, device=gpu)
    ops.call('dv_gemm_f32', K, N, M, xg, 1, K, (xg @ wg[:, :K].t()).contiguous(), N, 1, acc, N, 0.5, 1)
    ref = 1.0 + 0.5 * (x.double().t() @ (x.double() @ w[:, :K].double().t()))
    assert float((acc.cpu().double() - ref).abs().max()) <= 5e-5 * float(ref.abs().max()) + 1e-6


def test_sgd_and_ema(gpu):
    n = 1003
    p0, g = rnd(n, seed=25), rnd(n, seed=26)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([p], lr=0.003, momentum=0.9, weight_decay=1e-4)
    pg, buf = torch.zeros(1008, device=gpu), torch.zeros(1008, device=gpu)
    pg[:n] = p0.to(gpu)
    gg = torch.zeros(1008, device=gpu)
    gg[:n] = g.to(gpu)
    cp = torch.zeros(1008, dtype=torch.bfloat16, device=gpu)
    for _ in range(3):
        p.grad = g.clone()
        opt.step()
        ops.call('dv_sgd_momentum', pg, gg, buf, n, 0.003, 0.9, 1e-4, 1.0, DV_BF16, cp)
    close(pg[:n], p.detach(), DV_F32, 'sgd', factor=5)
    close(cp[:n], p.detach().to(torch.bfloat16).float(), DV_F32, 'sgd bf16 copy', factor=500)
    k, qq = rnd(n, seed=27), rnd(n, seed=28)
    kg = k.to(gpu)
    ops.call('dv_ema', kg, qq.to(gpu), n, 0.999, DV_F32, None)
    close(kg, k * 0.999 + qq * (1 - 0.999), DV_F32, 'ema')


def test_classifier_head_ops(gpu):
    """softmax cross-entropy (loss, gradient, rank of the target), inverted dropout, train-mode BatchNorm1d and the arena
    Linear against torch on the same numbers"""
    from dualvar_amd import functional as DF
    from dualvar_amd.engine import ParamStore
    R, K, Fd = 6, 101, 64
    lg = (3 * rnd(R, K, seed=81)).to(gpu).requires_grad_(True)
    tgt = torch.tensor([5, 100, 0, 17, 17, 42], device=gpu)
    loss, rank0 = DF.cross_entropy(lg, tgt)
    (2.0 * loss).backward()
    lr = lg.detach().cpu().clone().requires_grad_(True)
    ref = F.cross_entropy(lr, tgt.cpu())
    (2.0 * ref).backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    close(lg.grad, lr.grad, DV_F32, 'ce grad')
    assert torch.equal(rank0.cpu().long(), (lr.detach() > lr.detach().gather(1, tgt.cpu()[:, None])).sum(1))
    # dropout: kept entries scaled by 1/(1-p), gradient through the same mask
    x = rnd(R, Fd, seed=82).to(gpu).requires_grad_(True)
    torch.manual_seed(3)
    y = DF.dropout(x, 0.5)
    keep = (y != 0)
    assert 0.2 < float(keep.float().mean()) < 0.8
    assert torch.allclose(y[keep], 2.0 * x.detach()[keep])
    y.sum().backward()
    assert torch.equal(x.grad != 0, keep) and torch.allclose(x.grad[keep], torch.full_like(x.grad[keep], 2.0))
    # BatchNorm1d (train) -> Linear through a ParamStore arena
    bn, lin = torch.nn.BatchNorm1d(Fd), torch.nn.Linear(Fd, 10)
    bn.weight.data.copy_(1 + 0.2 * rnd(Fd, seed=83))
    bn.bias.data.copy_(0.1 * rnd(Fd, seed=84))
    bn_r, lin_r = torch.nn.BatchNorm1d(Fd), torch.nn.Linear(Fd, 10)
    bn_r.load_state_dict(bn.state_dict())
    lin_r.load_state_dict(lin.state_dict())
    st = ParamStore()
    st.add_bn(bn)
    st.add_conv(lin.weight, need_dgrad=False)
    st.add_vec(lin.bias)
    st.materialize(gpu, DV_F32)
    xi = (2 * rnd(R, Fd, seed=85) + 0.3)
    xg = xi.to(gpu).requires_grad_(True)
    out = DF.linear(DF.batchnorm1d_train(xg, st, bn), st, lin)
    gy = rnd(R, 10, seed=86)
    out.backward(gy.to(gpu))
    xr = xi.clone().requires_grad_(True)
    out_r = lin_r(bn_r.train()(xr))
    out_r.backward(gy)
    close(out, out_r, DV_F32, 'bn1d+linear fwd', factor=5)
    close(xg.grad, xr.grad, DV_F32, 'bn1d+linear dx', factor=20)
    close(lin.weight.grad, lin_r.weight.grad, DV_F32, 'linear dW', factor=5)
    close(lin.bias.grad, lin_r.bias.grad, DV_F32, 'linear db', factor=5)
    close(bn.weight.grad, bn_r.weight.grad, DV_F32, 'bn1d dgamma', factor=20)
    close(bn.bias.grad, bn_r.bias.grad, DV_F32, 'bn1d dbeta', factor=5)
    close(bn.running_mean, bn_r.running_mean, DV_F32, 'bn1d running mean', factor=5)
    close(bn.running_var, bn_r.running_var, DV_F32, 'bn1d running var', factor=5)


def test_nn_retrieval_matches_the_reference_recipe(gpu):
    """dualvar_amd.utils.retrieval == classifier.py:964-981 (centre, normalise, matmul, topk hit-rate) in torch"""
    from dualvar_amd.utils.retrieval import nn_retrieval, video_features
    g = torch.Generator().manual_seed(91)
    ncls, D = 7, 64
    proto = torch.randn(ncls, D, generator=g)
    ytr, yte = torch.randint(0, ncls, (90,), generator=g), torch.randint(0, ncls, (40,), generator=g)
    clips_tr = proto[ytr].repeat_interleave(10, 0) + 9.0 * torch.randn(900, D, generator=g) + 0.7
    clips_te = proto[yte].repeat_interleave(10, 0) + 9.0 * torch.randn(400, D, generator=g) + 0.7
    ftr, fte = video_features(clips_tr.to(gpu), 10), video_features(clips_te.to(gpu), 10)
    assert torch.allclose(ftr.cpu(), clips_tr.view(90, 10, D).mean(1), atol=1e-5)
    acc, sim = nn_retrieval(fte, yte, ftr, ytr)
    a, b = clips_te.view(40, 10, D).mean(1), clips_tr.view(90, 10, D).mean(1)
    a, b = F.normalize(a - a.mean(0, keepdim=True), dim=1), F.normalize(b - b.mean(0, keepdim=True), dim=1)
    ref = a @ b.t()
    close(sim, ref, DV_F32, 'retrieval similarity', factor=10)
    for k in (1, 5, 10, 20, 50):
        idx = ref.topk(k, dim=1).indices
        want = float((ytr[idx] == yte[:, None]).any(1).float().mean())
        assert abs(acc[k] - want) < 1e-6, (k, acc[k], want)
    assert 0.2 < acc[1] < 1.0 and acc[50] >= acc[1]           # a non-trivial case: neither chance nor saturated


# ---------------------------------------------------------------------------------------------------------------
# fp8 pointwise path (BASELINE configs[4])
FP8_T = {0: (torch.float8_e4m3fn, 448.0), 1: (torch.float8_e5m2, 57344.0)}


@pytest.mark.parametrize('fmt', [0, 1])
@pytest.mark.parametrize('dtype', [DV_BF16, DV_F32])
def test_quantize_fp8_against_torch(gpu, dtype, fmt):
    """dv_quantize_fp8: per-tensor scale = amax / FMAX, q = fp8_rne(x / scale), byte for byte torch's OCP float8 cast"""
    tdt = ops.TORCH_DTYPE[dtype]
    M, Cn, ld = 700, 48, 80
    wide = (rnd(M, ld, seed=21) * 3).to(tdt).to(gpu)
    x = wide[:, 16:16 + Cn]                                                  # a channel slice of a wider buffer
    q, scale = ops.quantize_fp8(x, M, Cn, ld, fmt, dtype)
    torch.cuda.synchronize()
    t8, fmax = FP8_T[fmt]
    amax = float(x.float().abs().max())
    assert abs(float(scale) - amax / fmax) <= 1e-6 * amax / fmax
    ref = (x.float() / scale).clamp(-fmax, fmax).to(t8)
    assert torch.equal(q.view(torch.uint8), ref.view(torch.uint8))
    z, sz = ops.quantize_fp8(torch.zeros(64, 16, dtype=tdt, device=gpu), 64, 16, 16, fmt, dtype)     # all-zero tensor: scale 1
    assert float(sz) == 1.0 and int(z.max()) == 0


@pytest.mark.parametrize('case', [('pw_96_80', 3, 96, 2, 20, 25, 80), ('pw_256_64', 2, 256, 4, 14, 14, 64), ('pw_64_256', 1, 64, 8, 28, 28, 256)],
                         ids=lambda c: c[0])
def test_pointwise_conv_fp8_fwd_dgrad(gpu, case):
    """dv_conv3d_fwd_fp8 / dv_conv3d_dgrad_fp8 against torch fp32 ON THE QUANTISED OPERANDS: the kernels are then exact up to
    fp32 accumulation order and the bf16 rounding of the output -- what fp8 itself costs is priced separately, below:
    tolerance statement of the fp8 mode = at most 2x the error e4m3 (e5m2 for gradients) rounding of the operands alone
    causes in the fp32 product (printed)."""
    name, N, Cin, T, H, W, Cout = case
    x = q(rnd(N, Cin, T, H, W, seed=1).relu(), DV_BF16)
    w = q(rnd(Cout, Cin, 1, 1, 1, seed=2, scale=Cin ** -0.5), DV_BF16)
    gy = q(rnd(N, Cout, T, H, W, seed=3), DV_BF16)
    xa = ops.act_from_ncdhw(x.to(gpu), DV_BF16)
    dya = ops.act_from_ncdhw(gy.to(gpu), DV_BF16)
    ya = ops.new_act(N, T, H, W, Cout, DV_BF16, gpu, zero=True)
    M = xa.rows
    assert xa.cpitch % 16 == 0 and ya.cpitch % 16 == 0
    x8, sx = ops.quantize_fp8(xa, M, xa.cpitch, xa.ld, 0, DV_BF16)
    wf = ops.pack_weight(w.to(gpu), xa.cpitch).to(torch.bfloat16).view(Cout, xa.cpitch)               # [Cout][CinP]
    w8, sw = ops.quantize_fp8(wf, Cout, xa.cpitch, xa.cpitch, 0, DV_BF16)
    d = ops.conv_desc(DV_BF16, xa, ya, (1, 1, 1), (1, 1, 1), (0, 0, 0), flags=ops.DV_STATS)
    d.ldx = x8.stride(0)
    tiles = ops.stat_tiles(d)
    stats = torch.zeros(2, Cout, tiles, device=gpu)
    ops.conv_fwd_fp8(d, x8, w8, sx, sw, ya, stats)
    torch.cuda.synchronize()
    xq = x8.view(torch.float8_e4m3fn).float() * sx
    wq = w8.view(torch.float8_e4m3fn).float() * sw
    ref = (xq @ wq.t())[:, :Cout]                                                                     # [M, Cout] fp32
    got = ya.buf[:, :Cout].float()
    exact = (xa.buf.float()[:, :xa.cpitch] @ wf.float().t())[:, :Cout]
    e_kernel = float((got - ref).abs().max() / ref.abs().max())
    e_fp8 = float((ref - exact).abs().max() / exact.abs().max())
    print(f'{name}: fwd kernel vs fp32-on-quantised {e_kernel:.2e} (bf16 output), fp8 quantisation itself {e_fp8:.2e}')
    assert e_kernel < 6e-3                                                                           # bf16 rounding of y
    assert float((got - exact).abs().max() / exact.abs().max()) < 2 * e_fp8 + 6e-3
    local = torch.zeros(2 * Cout + 1, device=gpu)
    ops.call('dv_bn_reduce_stats', stats, tiles, ops.tile_rows(d), Cout, M, Cout, local)
    close(local[:Cout] / M, got.mean(0), DV_F32, name + ' mean', factor=5)
    # data gradient: dy e5m2 x w e4m3 (dgrad layout [Cin][CoutP])
    dy8, sdy = ops.quantize_fp8(dya, M, dya.cpitch, dya.ld, 1, DV_BF16)
    wd = torch.zeros(Cin, dya.cpitch, device=gpu, dtype=torch.bfloat16)
    wd[:, :Cout] = w.view(Cout, Cin).t().to(gpu).to(torch.bfloat16)
    wd8, swd = ops.quantize_fp8(wd, Cin, dya.cpitch, dya.cpitch, 0, DV_BF16)
    dxa = xa.like()
    dd = ops.conv_desc(DV_BF16, xa, dya, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    dd.ldy = dy8.stride(0)
    ops.conv_dgrad_fp8(dd, dy8, wd8, sdy, swd, dxa)
    torch.cuda.synchronize()
    ref = ((dy8.view(torch.float8_e5m2).float() * sdy) @ (wd8.view(torch.float8_e4m3fn).float() * swd).t())[:, :Cin]
    gotd = dxa.buf[:, :Cin].float()
    e_k = float((gotd - ref).abs().max() / ref.abs().max())
    print(f'{name}: dgrad kernel vs fp32-on-quantised {e_k:.2e}')
    assert e_k < 6e-3
    # accumulate form
    ddacc = ops.conv_desc(DV_BF16, xa, dya, (1, 1, 1), (1, 1, 1), (0, 0, 0), flags=ops.DV_ACCUM)
    ddacc.ldy = dy8.stride(0)
    ops.conv_dgrad_fp8(ddacc, dy8, wd8, sdy, swd, dxa)
    assert float((dxa.buf[:, :Cin].float() - 2 * ref).abs().max() / ref.abs().max()) < 1.5e-2


@pytest.mark.parametrize('dtype', [DV_F32, DV_BF16])
@pytest.mark.parametrize('k,s,p', [((3, 3, 3), (1, 1, 1), (1, 1, 1)), ((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1))])
def test_maxpool_on_channel_slices(gpu, dtype, k, s, p):
    """input, output and both gradients as channel slices of wider buffers (pitch > C, offset > 0) -- the concat-by-slice
    layout of the Inception blocks -- through the staged (3x3x3 / 1), quad (3x3 / 2) and gather kernels: the neighbours of
    the slices stay untouched"""
    N, C_, T, H, W = 2, 40, 4, 13, 14
    x = F.relu(q(rnd(N, C_, T, H, W, seed=91), dtype))
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, k, s, p)
    gy = q(rnd(*yr.shape, seed=92), dtype)
    yr.backward(gy)
    To, Ho, Wo = yr.shape[2:]
    wide_x = ops.new_act(N, T, H, W, C_ + 32, dtype, gpu)
    wide_x.buf.fill_(5.0)
    xs = wide_x.slice(16, C_)
    xs_src = ops.act_from_ncdhw(x.to(gpu), dtype)
    wide_x.buf[:, 16:16 + xs_src.buf.shape[1]] = xs_src.buf
    # (the descriptor carries ONE pitch per side: dx has x's pitch, dy has y's)
    wide_y, wide_dy = (ops.new_act(N, To, Ho, Wo, C_ + 24, dtype, gpu) for _ in range(2))
    wide_dx = ops.new_act(N, T, H, W, C_ + 32, dtype, gpu)
    for wbuf in (wide_y, wide_dx):
        wbuf.buf.fill_(7.0)
    ys, dys, dxs = wide_y.slice(8, C_), wide_dy.slice(8, C_), wide_dx.slice(16, C_)
    g_src = ops.act_from_ncdhw(gy.to(gpu), dtype)
    wide_dy.buf.fill_(3.0)
    wide_dy.buf[:, 8:8 + g_src.buf.shape[1]] = g_src.buf
    idx = torch.zeros(ys.rows, ops.cp8(C_), dtype=torch.uint8, device=gpu)
    d = ops.pool_desc(dtype, xs, ys, k, s, p)
    ops.call('dv_maxpool3d_fwd', d, xs, ys, idx)
    got = wide_y.buf[:, 8:8 + C_].float().view(N, To, Ho, Wo, C_).permute(0, 4, 1, 2, 3)
    assert torch.equal(got.cpu(), yr.detach())
    assert float((wide_y.buf[:, :8].float() - 7).abs().max()) == 0 and float((wide_y.buf[:, 8 + ops.cp8(C_):].float() - 7).abs().max()) == 0
    ops.call('dv_maxpool3d_bwd', d, dys, idx, dxs, 0)
    gotx = wide_dx.buf[:, 16:16 + C_].float().view(N, T, H, W, C_).permute(0, 4, 1, 2, 3)
    close(gotx, q(xr.grad, dtype), dtype, 'pool bwd on slices')
    assert float((wide_dx.buf[:, :16].float() - 7).abs().max()) == 0 and float((wide_dx.buf[:, 16 + ops.cp8(C_):].float() - 7).abs().max()) == 0


def test_random_shapes_conv_and_pool(gpu):
    """tools/fuzz_ops.py with a fixed seed: 60 random (shape, window, stride, dtype) cases through conv fwd / dgrad / wgrad and
    max-pool fwd / bwd against PyTorch on the CPU -- odd channel counts, planes that are not a multiple of the 32-pixel chunk
    of the weight gradient's t-inner row order, ragged pool tiles"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_ops.py'), '--n', '60', '--seed', '7'], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and 'ok: 60 random cases' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


BIG_M_CASES = [
    # name, N, Cin, T, H, W, Cout, k, s, p   -- >= 200 704 output rows behind a 64-column tile (dgrad: input rows behind Cin = 64)
    ('sp3_200k', 8, 64, 8, 56, 56, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('tm7_s2_200k', 16, 64, 8, 56, 56, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0)),
    # R(2+1)D's conv2 block at its own width (backbone/r21d.py:47-49: 144 mid channels), 8 frames of 56 x 56
    ('r21d_sp3_144', 6, 64, 8, 56, 56, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r21d_tm3_144', 8, 144, 8, 56, 56, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    # S3D-G's Conv_2c pair on four-frame maps (the temporal form of the LDS-staged kernel with T = 4)
    ('c2c_tm3_t4', 20, 64, 4, 56, 56, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
]
# which of them dv_conv3d_fwd / the stride-1 dv_conv3d_dgrad run on the LDS-staged input-tile kernel (csrc/conv_tap.hip): 1 spatial, 2 temporal
BIG_M_TAP = {'sp3_200k': 1, 'tm7_s2_200k': 0, 'r21d_sp3_144': 1, 'r21d_tm3_144': 2, 'c2c_tm3_t4': 2}


@pytest.mark.parametrize('case', BIG_M_CASES, ids=[c[0] for c in BIG_M_CASES])
def test_fp32_conv_at_headline_tile_sizes_against_cpu_conv3d(gpu, case):
    """The fp32 instantiations only big layers reach, against an oracle: M >= 200 k rows x 64 channels selects
    conv_gemm<f32,FWD,256,64> / <f32,DGRAD,256,64> (pick_tile: >= 512 tiles of 256 rows) -- or, for the stride-1 "same" 1x3x3 /
    3x1x1 cases, the LDS-staged input-tile kernel conv_tap<f32,*,sp|tm,256,64> (round 4; asserted by dv_conv3d_tap_kind) -- and a
    weight gradient of ~100 row splits in the t-inner row order (7x1x1, 3x1x1) -- at the 8..12-clip sizes of the other tests these
    layers run 64- / 128-row tiles and a handful of splits.  Oracle: torch.nn.functional.conv3d on the CPU in float64 (the op the reference's nn.Conv3d is,
    backbone/s3dg.py:39-42); the CPU fp32 result measures what fp32 arithmetic itself costs, the kernels must stay within 4x that."""
    from dualvar_amd._lib import DV_W3
    import ctypes as C
    from dualvar_amd import _lib as L
    if _EXACT:
        pytest.skip('the pre-split-weight instantiations do not exist under DUALVAR_F32_EXACT=1')
    name, N, Cin, T, H, W, Cout, k, s, p = case
    x = rnd(N, Cin, T, H, W, seed=1).relu_()
    w = rnd(Cout, Cin, *k, seed=2, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, s, p)
    gy = rnd(*yr.shape, seed=3)
    yr.backward(gy.double())
    x32, w32 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y32 = F.conv3d(x32, w32, None, s, p)
    y32.backward(gy)

    def rel(a, ref):
        return float((a.double().cpu() - ref).abs().max() / ref.abs().max())
    cpu = {'fwd': rel(y32.detach(), yr.detach()), 'dgrad': rel(x32.grad, xr.grad), 'wgrad': rel(w32.grad, wr.grad)}

    xa = ops.act_from_ncdhw(x.to(gpu), DV_F32)
    To, Ho, Wo = yr.shape[2:]
    ya = ops.new_act(N, To, Ho, Wo, Cout, DV_F32, gpu, zero=True)
    d = ops.conv_desc(DV_F32, xa, ya, k, s, p, flags=ops.DV_STATS | DV_W3)
    rows_, cols_ = C.c_int32(), C.c_int32()
    lib = L.load()
    for dg in (0, 1):
        assert lib.dv_conv3d_tile_shape(C.byref(d), dg, C.byref(rows_), C.byref(cols_)) == 0
        if not (dg and max(s) > 1) and Cout == 64:       # (a strided data gradient runs one launch per parity class on fewer rows each)
            assert (rows_.value, cols_.value) == (256, 64), (name, dg, rows_.value, cols_.value)
    assert lib.dv_conv3d_tap_kind(C.byref(d), 0) == BIG_M_TAP[name], name
    assert ops.tile_rows(d) == 256, name
    wp = ops.pack_weight(w.to(gpu), ops.cp8(Cin))
    tiles = ops.stat_tiles(d)
    stats = torch.zeros(2, Cout, tiles, device=gpu)
    ops.conv_fwd(d, xa, ops.pack_w3(wp.view(Cout, -1)), None, ya, stats)
    got = {'fwd': rel(ops.act_to_ncdhw(ya), yr.detach())}
    # fused BatchNorm partials of the 256-row tiles == statistics of the float64 output
    M = N * To * Ho * Wo
    local = torch.zeros(2 * Cout + 1, device=gpu)
    ops.call('dv_bn_reduce_stats', stats, tiles, ops.tile_rows(d), Cout, M, Cout, local)
    mean_ref, var_ref = yr.detach().mean(dim=(0, 2, 3, 4)), yr.detach().var(dim=(0, 2, 3, 4), unbiased=False)
    assert float(((local[:Cout].cpu().double() / M) - mean_ref).abs().max()) <= 2e-6 * float(var_ref.sqrt().max())
    assert torch.allclose(local[Cout:2 * Cout].cpu().double() / M, var_ref, rtol=5e-6)

    dya = ops.act_from_ncdhw(gy.to(gpu), DV_F32)
    d2 = ops.conv_desc(DV_F32, xa, dya, k, s, p)
    r_, c_, sp_ = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.dv_conv3d_wgrad_tile(C.byref(d2), C.byref(r_), C.byref(c_), C.byref(sp_)) == 0
    assert sp_.value >= 32, (name, sp_.value)
    dw = torch.zeros_like(wp)
    ops.conv_wgrad(d2, xa, dya, dw)
    dw2 = torch.zeros_like(wp)
    ops.conv_wgrad(d2, xa, dya, dw2)
    assert torch.equal(dw, dw2)
    got['wgrad'] = rel(ops.unpack_weight(dw, w.shape), wr.grad)

    taps = k[0] * k[1] * k[2]
    wd = torch.zeros(Cin, taps, ops.cp8(Cout), device=gpu)
    wd[:, :, :Cout] = w.to(gpu).reshape(Cout, Cin, taps).permute(1, 2, 0)
    dxa = ops.new_act(N, T, H, W, Cin, DV_F32, gpu, zero=True)
    if max(s) == 1:
        dg3 = ops.conv_desc(DV_F32, xa, dya, k, s, p, flags=DV_W3)
        assert lib.dv_conv3d_tap_kind(C.byref(dg3), 1) == BIG_M_TAP[name], name
        ops.conv_dgrad(dg3, dya, ops.pack_w3(wd.view(Cin, -1)), dxa)
    else:
        # t-strided (the 7x1x1 / stride-2 stem conv): with pre-split weights every parity class runs on the temporal form of the
        # LDS-staged kernel; without them on conv_gemm -- both against float64
        dg3 = ops.conv_desc(DV_F32, xa, dya, k, s, p, flags=DV_W3)
        assert lib.dv_conv3d_tap_kind(C.byref(dg3), 1) == 2, name
        ops.conv_dgrad(dg3, dya, ops.pack_w3(wd.view(Cin, -1)), dxa)
        got['dgrad_gemm'] = rel(ops.act_to_ncdhw(dxa), xr.grad)
        dxa.buf.zero_()
        ops.conv_dgrad(d2, dya, wd, dxa)
        got['dgrad'], got['dgrad_gemm'] = got['dgrad_gemm'], rel(ops.act_to_ncdhw(dxa), xr.grad)
        cpu['dgrad_gemm'] = cpu['dgrad']
    got['dgrad'] = got.get('dgrad', rel(ops.act_to_ncdhw(dxa), xr.grad))
    print(name, 'relative-to-max error vs float64:', {k_: ('%.2e (cpu fp32 %.2e)' % (got[k_], cpu[k_])) for k_ in got},
          'wgrad splits', sp_.value)
    for k_ in got:
        assert got[k_] <= max(4 * cpu[k_], 2e-6), (name, k_, got[k_], cpu[k_])


def test_lds_staged_input_tile_kernel_on_small_and_ragged_shapes(gpu):
    """csrc/conv_tap.hip takes problems of >= 512 tiles; DUALVAR_CONV_TAP_GRID=1 (read once per process, hence the child) forces
    tools/tap_check.py's small shapes onto it: partial last tiles, tiles that straddle planes and clips, 5 x 3 .. 56 x 56 planes,
    T = 2 / 4 / 8, odd channel counts; forward with the BatchNorm partials, data gradient and `+=`, all against torch's conv3d
    in float64, and a sentinel behind the output."""
    import os
    import subprocess
    import sys
    if _EXACT:
        pytest.skip('the LDS-staged kernels belong to the split mode (DUALVAR_F32_EXACT=1 runs none of them)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DUALVAR_CONV_TAP_GRID='1')
    for bm128 in ('0', '1'):             # the 256-row tiles, then the 128-row tiles small launches get by default (tap_bm)
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'tap_check.py')], capture_output=True, text=True, timeout=600,
                           env=dict(env, DUALVAR_CONV_TAP_BM128=bm128))
        assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout[-3000:] + r.stderr[-2000:]
        assert 'kind fwd 0' not in r.stdout and 'dgrad 0' not in r.stdout, r.stdout[-3000:]
        assert ('rows 128' in r.stdout) == (bm128 == '1') and 'rows 256' in r.stdout, r.stdout[-3000:]     # (eight-frame temporal tiles stay 256)
    # the pixel-pair stem form of the LDS-staged weight gradient (conv_wgrad_pp_kernel, with and without the BatchNorm apply
    # inside) on the small stem shapes of this file: the same tests, with the size thresholds lifted
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k',
                        '(batchnorm_backward_apply_inside and shape0) or rgb_stem_as_pixel_pair or (conv_fwd_dgrad_wgrad and pair_stem)'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=root)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0 and ' passed' in tail, tail


SENTINEL_CASES = [
    # name, N, Cin, T, H, W, Cout: pointwise convs whose ROW COUNT is not a multiple of the tile height (nor of 32): the last tile
    # is partial.  256 x 64 tiles (>= 512 tiles of 256 rows), 128 x 128 tiles (>= 1 024 tiles of 128 rows), 64-row tiles.
    ('t256', 1, 16, 3, 121, 121, 192),
    ('t128', 1, 16, 3, 211, 211, 128),
    ('t64', 1, 32, 1, 37, 27, 96),
    ('t64_c40', 1, 16, 1, 33, 31, 40),
]


@pytest.mark.parametrize('case', SENTINEL_CASES, ids=[c[0] for c in SENTINEL_CASES])
def test_fp32_partial_tiles_do_not_write_behind_the_output(gpu, case):
    """the register-direct fp32 epilogue put the row term of its store addresses into soffset, which the
    buffer range check does not cover, so a partial last tile stored (zeros) up to 31+ rows BEHIND the tensor.  Forward, data
    gradient and `+=` data gradient with M % 32 != 0 into a view of a larger buffer whose trailing rows hold a sentinel:
    the sentinel survives, the values are right."""
    from dualvar_amd._lib import DV_W3
    name, N, Cin, T, H, W, Cout = case
    k, s_, p_ = (1, 1, 1), (1, 1, 1), (0, 0, 0)
    M = N * T * H * W
    assert M % 32 != 0
    x = rnd(N, Cin, T, H, W, seed=1)
    w = rnd(Cout, Cin, 1, 1, 1, seed=2, scale=Cin ** -0.5)
    yr = F.conv3d(x, w)
    gy = rnd(*yr.shape, seed=3)
    dxr = F.conv_transpose3d(gy, w)
    SENT, EXTRA = 12345.0, 300

    def guarded(C_):
        ld = ops.cp8(C_)
        buf = torch.full((M + EXTRA, ld), SENT, dtype=torch.float32, device=gpu)
        buf[:M] = 0
        return ops.Act(buf[:M], N, T, H, W, C_, ld, 0, DV_F32, ld), buf
    xa = ops.act_from_ncdhw(x.to(gpu), DV_F32)
    wp = ops.pack_weight(w.to(gpu), ops.cp8(Cin))
    wd = torch.zeros(Cin, 1, ops.cp8(Cout), device=gpu)
    wd[:, :, :Cout] = w.to(gpu).reshape(Cout, Cin, 1).permute(1, 2, 0)
    for w3 in ((False, True) if not _EXACT else (False,)):
        ya, ybuf = guarded(Cout)
        d = ops.conv_desc(DV_F32, xa, ya, k, s_, p_, flags=DV_W3 if w3 else 0)
        ops.conv_fwd(d, xa, ops.pack_w3(wp.view(Cout, -1)) if w3 else wp, None, ya, None)
        torch.cuda.synchronize()
        assert bool((ybuf[M:] == SENT).all()), (name, w3, 'forward wrote behind row M')
        close(ops.act_to_ncdhw(ya), yr, DV_F32, name + ' fwd')
        dya = ops.act_from_ncdhw(gy.to(gpu), DV_F32)
        dxa, dbuf = guarded(Cin)
        for flags in (0, ops.DV_ACCUM):
            d2 = ops.conv_desc(DV_F32, dxa, dya, k, s_, p_, flags=flags | (DV_W3 if w3 else 0))
            ops.conv_dgrad(d2, dya, ops.pack_w3(wd.view(Cin, -1)) if w3 else wd, dxa)
            torch.cuda.synchronize()
            assert bool((dbuf[M:] == SENT).all()), (name, w3, flags, 'data gradient wrote behind row M')
        close(ops.act_to_ncdhw(dxa), 2 * dxr, DV_F32, name + ' dgrad, then +=', factor=2)


@pytest.mark.parametrize('cfg', [0, 2])
def test_second_fp32_weight_gradient_form_stays_correct(gpu, cfg):
    """conv_wgrad_f32s_kernel is opt-in (DUALVAR_WGRAD_F32S, read once per process: DESIGN section 4 explains why it is not the
    default).  The weight-gradient op tests of this file under it, in a child process: 64 x 256 and 128 x 128 tiles."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DUALVAR_WGRAD_F32S=str(cfg))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k',
                        'conv_wgrad_row_splits or (conv_fwd_dgrad_wgrad and (sp3 or tm3 or full3 or big_n))'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=500,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0 and ' passed' in tail, tail


BN_IN_CASES = [
    # name, N, Cin, cin pitch, T, H, W, Cout, k, s, p, relu, expected dv_conv3d_bn_in_ok
    ('c2c_tm3_t4', 20, 64, 64, 4, 56, 56, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), True, 1),
    ('stem_tm7_s2', 16, 64, 64, 8, 56, 56, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), True, 2),
    ('stem_tm7_s2_ragged', 11, 64, 64, 8, 56, 55, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), True, 2),        # 529.4 tiles of 256 rows
    ('odd_tm3_t2', 12, 83, 96, 2, 28, 28, 144, (3, 1, 1), (1, 1, 1), (1, 0, 0), True, 1),
    ('tm3_t2_linear', 12, 48, 48, 2, 28, 28, 136, (3, 1, 1), (1, 1, 1), (1, 0, 0), False, 1),
    ('r21d_tm3_t8', 8, 144, 144, 8, 56, 56, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), True, 0),        # 8 frames: no staged weight gradient
    ('sp3', 8, 64, 64, 4, 56, 56, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), True, 0),
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', BN_IN_CASES, ids=[c[0] for c in BN_IN_CASES])
def test_batchnorm_on_load_gives_the_bits_of_apply_then_conv(gpu, case):
    """dv_conv3d_fwd_bn_in / dv_conv3d_wgrad_bn_in (conv -> BN -> ReLU -> conv with the BatchNorm's output never written:
    backbone/s3dg.py:30-65 STConv3d, and the stem pair s3dg.py:151) against the two-launch plan dv_bn_apply -> dv_conv3d_fwd /
    dv_conv3d_wgrad on the same inputs: the same expression feeds the same kernels' products, so y, the BatchNorm partials and dW
    must agree BIT FOR BIT; where dv_conv3d_bn_in_ok says 0 both entry points refuse."""
    import ctypes as C
    from dualvar_amd import _lib as L
    from dualvar_amd._lib import DV_W3, DV_RELU
    if _EXACT:
        pytest.skip('the pre-split-weight instantiations do not exist under DUALVAR_F32_EXACT=1')
    name, N, Cin, cpitch, T, H, W, Cout, k, s, p, relu, want = case
    lib = L.load()
    xr = ops.act_from_ncdhw(rnd(N, Cin, T, H, W, seed=11).to(gpu), DV_F32, cpitch=cpitch)
    CP = ops.cp8(Cin)
    scale, shift = torch.zeros(CP, device=gpu), torch.zeros(CP, device=gpu)
    scale[:Cin] = (rnd(Cin, seed=12) * 0.5 + 1.0).to(gpu)
    shift[:Cin] = (rnd(Cin, seed=13) * 0.3).to(gpu)
    ya = ops.new_act(N, T, H, W, Cin, DV_F32, gpu, cpitch=cpitch, zero=True)
    M_in = N * T * H * W
    ops.call('dv_bn_apply', DV_F32, xr, xr.ld, scale, shift, None, 0, ya, ya.ld, M_in, Cin, DV_RELU if relu else 0)
    To, Ho, Wo = ops.conv_out_dims(xr, k, s, p)
    w = rnd(Cout, Cin, *k, seed=14, scale=(Cin * k[0]) ** -0.5)
    wp = ops.pack_weight(w.to(gpu), cpitch)
    w3 = ops.pack_w3(wp.view(Cout, -1))
    out1 = ops.new_act(N, To, Ho, Wo, Cout, DV_F32, gpu, zero=True)
    out2 = ops.new_act(N, To, Ho, Wo, Cout, DV_F32, gpu, zero=True)
    d = ops.conv_desc(DV_F32, ya, out1, k, s, p, flags=ops.DV_STATS | DV_W3)
    assert int(lib.dv_conv3d_bn_in_ok(C.byref(d))) == want, name
    tiles = ops.stat_tiles(d)
    st1, st2 = torch.zeros(2, Cout, tiles, device=gpu), torch.zeros(2, Cout, tiles, device=gpu)
    bn = ops.bn_in_desc(scale, shift, relu)
    dya = ops.act_from_ncdhw(rnd(N, Cout, To, Ho, Wo, seed=15).to(gpu), DV_F32)
    d2 = ops.conv_desc(DV_F32, ya, dya, k, s, p)
    if not want:
        if name != 'r21d_tm3_t8':        # (that one's forward alone could: the pair is refused because its weight gradient cannot)
            with pytest.raises(L.DualVarHipError, match='DV_EUNSUPPORTED'):
                ops.conv_fwd_bn_in(d, xr, bn, w3, out2, st2)
        with pytest.raises(L.DualVarHipError, match='DV_EUNSUPPORTED'):
            ops.conv_wgrad_bn_in(d2, xr, bn, dya, torch.zeros_like(wp))
        return
    ops.conv_fwd(d, ya, w3, None, out1, st1)
    ops.conv_fwd_bn_in(d, xr, bn, w3, out2, st2)
    assert torch.equal(out1.buf, out2.buf), (name, float((out1.buf - out2.buf).abs().max()))
    assert torch.equal(st1, st2), name
    assert float(out1.buf.abs().max()) > 0.1
    dw1, dw2 = torch.zeros_like(wp), torch.zeros_like(wp)
    ops.conv_wgrad(d2, ya, dya, dw1)
    ops.conv_wgrad_bn_in(d2, xr, bn, dya, dw2)
    assert torch.equal(dw1, dw2), (name, float((dw1 - dw2).abs().max()))
    assert float(dw1.abs().max()) > 0.1


@pytest.mark.gpu
def test_pixel_pair_stem_forward_at_headline_line_length(gpu):
    """conv_pp_fwd_kernel (csrc/conv_tap.hip): the RGB stem conv (backbone/s3dg.py:151 Conv_1a.conv1, 1x7x7 / stride 2 / padding 3)
    on 40 frames of 112 x 112 -- 560 tiles of four 56-pixel output lines, the form the headline step runs -- against torch's conv3d
    in float64, with the BatchNorm partials of the 224-row tiles; the kernel must stay within 4x of what fp32 arithmetic itself
    costs on the CPU."""
    import ctypes as C
    import os
    import sys
    from dualvar_amd import _lib as L
    if _EXACT:
        pytest.skip('the pre-split-weight kernels do not exist under DUALVAR_F32_EXACT=1')
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import tap_check
    N, T, H, W, O = 5, 8, 112, 112, 64
    e = tap_check.check_pp_stem_forward(L.load(), gpu, N, T, H, W, O, 224)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, 3, T, H, W, generator=g)
    w = 0.2 * torch.randn(O, 3, 1, 7, 7, generator=g)
    y64 = F.conv3d(x.double(), w.double(), None, (1, 2, 2), (0, 3, 3))
    cpu = float((F.conv3d(x, w, None, (1, 2, 2), (0, 3, 3)).double() - y64).abs().max() / y64.abs().max())
    print('relative-to-max error vs float64: %.2e (cpu fp32 %.2e)' % (e, cpu))
    assert e <= max(4 * cpu, 2e-6), (e, cpu)
