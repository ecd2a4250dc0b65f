"""Size-independent properties at the BASELINE sizes (128 clips of 8x112x112 per GPU; fp32 -- the headline arithmetic -- and
bf16) -- where the CPU oracle would take minutes per layer, the kernels are checked through identities that hold at any size:

  * conv: <conv(x, w), dy> == <x, dgrad(dy, w)> == <w, wgrad(x, dy)>  (fwd / dgrad / wgrad are mutual adjoints) on the
    largest S3D-G layers, and exact homogeneity conv(x, 2w) == 2 conv(x, w) (a power-of-two scale is exact in bf16);
  * BatchNorm: output statistics (mean beta, variance gamma^2), sum(dx) == 0 and sum(dx * xhat) == 0 in the backward;
  * max-pool: every output is the maximum of its window maxima bound, gradient mass is conserved (sum dx == sum dy);
  * the whole step: a permutation of the batch permutes the logits (the statistics are batch-symmetric) and the loss of a
    full-size step equals the loss formula evaluated on the CPU from the returned logits.
All calls go through the C ABI (dualvar_amd.ops -> libdualvar_hip.so)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dualvar_amd import ops  # noqa: E402
from dualvar_amd.ops import DV_BF16, DV_F32  # noqa: E402

DTYPES = [pytest.param(DV_F32, id='fp32'), pytest.param(DV_BF16, id='bf16')]
from dualvar_amd import _lib as _L  # noqa: E402
_EXACT = _L.f32_exact()

N_CLIPS = 128

# (name, clips, T, H, W, Cin, Cout, k, s, p): the largest S3D-G layers at 128 clips of 8 x 112 x 112 and
# one stride-2 case; the same stem / Conv_2c layers at 16-frame clips (BASELINE configs[1]: S3D-G bf16, batch 64 x 2 views,
# 16 x 112 x 112); and three of the largest layers of the 2D3D-ResNet-50 on 32 x 224 x 224 clips (configs[4]) at batch 4 x 2
BIG_LAYERS = [
    ('Conv_2c.conv1 1x3x3', 128, 4, 28, 28, 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('Conv_2c.conv2 3x1x1', 128, 4, 28, 28, 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('Conv_1a.conv2 7x1x1 s2', 128, 8, 56, 56, 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0)),
    ('Mixed_3c entry 1x1x1', 128, 4, 14, 14, 256, 288, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('T16 Conv_1a.conv2 7x1x1 s2', 128, 16, 56, 56, 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0)),
    ('T16 Conv_2c.conv1 1x3x3', 128, 8, 28, 28, 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('T16 Mixed_4b branch1 3x1x1', 128, 4, 7, 7, 208, 208, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    # R(2+1)D (BASELINE configs[2]; backbone/r21d.py:47-49,176-266) at its own per-GPU size, 32 samples x 3 views of 8 x 112 x 112:
    # the conv2 block's 144 mid channels (52 % of the net's FLOPs), and conv3's strided 1x3x3 into 230 mid channels
    ('r21d conv2 spatial 1x3x3 64->144', 96, 8, 56, 56, 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r21d conv2 temporal 3x1x1 144->64', 96, 8, 56, 56, 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('r21d conv3 spatial 1x3x3 s2 64->230', 96, 8, 56, 56, 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ('r50@224 layer1 conv2 1x3x3', 8, 16, 56, 56, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r50@224 layer1 conv3 1x1x1', 8, 16, 56, 56, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50@224 layer3 conv1 3x1x1', 8, 16, 14, 14, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('r50@224 layer2 downsample 1x1x1 s(1,2,2)', 8, 16, 56, 56, 256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
]


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('layer', BIG_LAYERS, ids=[c[0] for c in BIG_LAYERS])
def test_conv_adjoint_identities_at_full_size(gpu, layer, dtype):
    """fp32: the instantiations only the headline-size step reaches -- conv_gemm<f32,*,256,64> (>= 131 k rows behind a 64-column
    tile), weight gradients with >= 100 row splits, the t-inner row order -- run here with the engine's pre-split weights
    (DV_W3) where the engine uses them (forward; stride-1 data gradient)."""
    from dualvar_amd._lib import DV_W3
    name, N_CLIPS, T, H, W, Ci, Co, k, s, p = layer
    tdt = ops.TORCH_DTYPE[dtype]
    f32 = dtype == DV_F32
    if f32 and _EXACT:
        pytest.skip('pre-split weights (DV_W3) do not exist under DUALVAR_F32_EXACT=1')
    g = torch.Generator(device='cpu').manual_seed(7)
    x = ops.new_act(N_CLIPS, T, H, W, Ci, dtype, gpu)
    x.buf.copy_(torch.randn(x.buf.shape, generator=g).relu_().to(tdt))
    To, Ho, Wo = ops.conv_out_dims(x, k, s, p)
    y, dy = ops.new_act(N_CLIPS, To, Ho, Wo, Co, dtype, gpu), ops.new_act(N_CLIPS, To, Ho, Wo, Co, dtype, gpu)
    dy.buf.copy_(torch.randn(dy.buf.shape, generator=g).to(tdt))
    dy.buf[:, Co:] = 0                                                                     # pad lanes [Co, cout_pitch) hold zeros
    taps = k[0] * k[1] * k[2]
    wf = (torch.randn(Co, taps, Ci, generator=g) * (taps * Ci) ** -0.5).to(gpu)            # master layout [Co][tap][Ci]
    w16 = wf.to(tdt)
    wd = torch.zeros(Ci, taps, ops.cp8(Co), dtype=tdt, device=gpu)                         # dgrad layout [Ci][tap][cout_pitch]
    wd[:, :, :Co] = w16.float().permute(2, 1, 0).to(tdt)
    d = ops.conv_desc(dtype, x, y, k, s, p)
    dx = x.like()
    if f32:
        d3 = ops.conv_desc(dtype, x, y, k, s, p, flags=DV_W3)
        ops.conv_fwd(d3, x, ops.pack_w3(w16.view(Co, -1)), None, y, None)
        import ctypes as C
        if max(s) == 1 or _L.load().dv_conv3d_tap_kind(C.byref(d3), 1):       # (t-strided stem conv: its parity classes on conv_tap)
            ops.conv_dgrad(d3, dy, ops.pack_w3(wd.view(Ci, -1)), dx)
        else:
            ops.conv_dgrad(d, dy, wd, dx)
    else:
        ops.conv_fwd(d, x, w16, None, y, None)
        ops.conv_dgrad(d, dy, wd, dx)
    dw = torch.zeros(Co, taps * Ci, device=gpu)
    ops.conv_wgrad(d, x, dy, dw)
    torch.cuda.synchronize()
    a = _dot(y.buf, dy.buf)                    # <conv(x,w), dy>   (y is bf16-rounded: 2^-9 relative per element)
    b = _dot(x.buf, dx.buf)                    # <x, dgrad(dy,w)>
    dw2 = torch.zeros(Co, taps * Ci, device=gpu)
    ops.conv_wgrad(d, x, dy, dw2)
    assert torch.equal(dw, dw2), 'the weight gradient must be reproducible bit for bit'
    c = _dot(w16.float().reshape(Co, -1), dw)  # <w, wgrad(x,dy)>  (fp32 accumulation, exact up to summation order)
    scale = float(y.buf.double().norm() * dy.buf.double().norm())
    print(f'{name}: <y,dy>={a:.6e} <x,dx>={b:.6e} <w,dw>={c:.6e} (|y||dy|={scale:.3e}) '
          f'|a-c|/scale={abs(a - c) / scale:.2e} |b-c|/scale={abs(b - c) / scale:.2e}')
    # bf16: rounding of y / dx to bf16 is unbiased: the inner products agree to ~2^-9 / sqrt(#elements) of |y||dy|
    # fp32: every element of y / dx / dw carries a relative error of a few 2^-24; the three inner products agree to 1e-6 of |y||dy|
    tol = 1e-6 if f32 else 2e-4
    assert abs(a - c) <= tol * scale and abs(b - c) <= tol * scale
    # homogeneity: doubling the weights doubles every output bit for bit (a power of two scales the bf16 triple of an fp32 weight
    # exactly as well)
    y2 = y.like()
    if f32:
        ops.conv_fwd(d3, x, ops.pack_w3((w16 * 2).view(Co, -1)), None, y2, None)
    else:
        ops.conv_fwd(d, x, (w16.float() * 2).to(torch.bfloat16), None, y2, None)
    assert torch.equal(y2.buf.float(), y.buf.float() * 2)


@pytest.mark.parametrize('dtype', DTYPES)
def test_batchnorm_statistics_and_backward_orthogonality_at_full_size(gpu, dtype):
    """the stem's BatchNorm (M = 128*8*56*56 = 3.2 M rows, C = 64): conv epilogue partials -> statistics -> apply ->
    backward reduce / apply; fp32 = the headline step's arithmetic (256 x 64 forward tiles, two-pass M2 partials)"""
    from dualvar_amd._lib import DV_NO_RELU_MASK, DV_W3
    f32 = dtype == DV_F32
    if f32 and _EXACT:
        pytest.skip('pre-split weights (DV_W3) do not exist under DUALVAR_F32_EXACT=1')
    tdt = ops.TORCH_DTYPE[dtype]
    T, H, W, C_ = 8, 56, 56, 64
    g = torch.Generator().manual_seed(11)
    xin = ops.new_act(N_CLIPS, T, H, W, 64, dtype, gpu)
    xin.buf.copy_(torch.randn(xin.buf.shape, generator=g).to(tdt))
    x = ops.new_act(N_CLIPS, T, H, W, C_, dtype, gpu)
    d = ops.conv_desc(dtype, xin, x, (1, 1, 1), (1, 1, 1), (0, 0, 0), flags=ops.DV_STATS | (DV_W3 if f32 else 0))
    w16 = (torch.randn(C_, 64, generator=g) / 8).to(gpu).to(tdt)
    tiles = ops.stat_tiles(d)
    part = torch.zeros(2, C_, tiles, device=gpu)
    ops.conv_fwd(d, xin, ops.pack_w3(w16) if f32 else w16, None, x, part)
    M = x.rows
    gamma, beta = (1 + 0.2 * torch.randn(C_, generator=g)).to(gpu), (0.1 * torch.randn(C_, generator=g)).to(gpu)
    local = torch.zeros(2 * C_ + 1, device=gpu)
    mean, invstd, scale, shift = (torch.zeros(C_, device=gpu) for _ in range(4))
    ops.call('dv_bn_stats_finalize', part, tiles, ops.tile_rows(d), C_, M, C_, local, gamma, beta, 1e-5, 0.1, None, None,
             mean, invstd, scale, shift)
    xs = x.buf.double()
    mean_ref, var_ref = xs.mean(0), xs.var(0, unbiased=False)
    # fused partials == direct statistics of the stored tensor (float64 reference)
    assert float((mean.double() - mean_ref).abs().max()) <= (2e-6 if f32 else 2e-5) * float(var_ref.sqrt().max())
    assert torch.allclose(invstd.double(), (var_ref + 1e-5).rsqrt(), rtol=2e-6 if f32 else 2e-4)
    y = x.like()
    ops.call('dv_bn_apply', dtype, x, x.ld, scale, shift, None, 0, y, y.ld, M, C_, 0)
    ys = y.buf.double()
    assert float((ys.mean(0) - beta.double()).abs().max()) < (2e-6 if f32 else 2e-3)          # (bf16: storage of y)
    # var(y) = gamma^2 var / (var + eps): the eps of the inverse standard deviation is part of the expected value
    want_std = gamma.abs().double() * (var_ref / (var_ref + 1e-5)).sqrt()
    assert float((ys.var(0, unbiased=False).sqrt() - want_std).abs().max()) < (2e-6 if f32 else 5e-3)
    dy = x.like()
    dy.buf.copy_(torch.randn(dy.buf.shape, generator=g).to(tdt))
    CP = ops.cp8(C_)
    sums = torch.zeros(8, 2, CP, device=gpu)
    ops.call('dv_bn_bwd_reduce', dtype, dy, dy.ld, y, y.ld, x, x.ld, mean, invstd, M, C_, DV_NO_RELU_MASK, sums, 8, None)
    dx = x.like()
    dgam, dbet = torch.zeros(C_, device=gpu), torch.zeros(C_, device=gpu)
    ops.call('dv_bn_bwd_apply', dtype, dy, dy.ld, y, y.ld, x, x.ld, mean, invstd, gamma, sums, 8, 1.0 / M, 1.0, dgam, dbet,
             dx, dx.ld, None, 0, M, C_, DV_NO_RELU_MASK)
    torch.cuda.synchronize()
    gs = dy.buf.double()
    assert torch.allclose(dbet.double(), gs.sum(0), rtol=1e-3, atol=1e-1)
    xhat = (xs - mean.double()) * invstd.double()
    assert torch.allclose(dgam.double(), (gs * xhat).sum(0), rtol=2e-3, atol=1.0)
    # the BatchNorm backward projects out the constant and the xhat direction of every channel
    dxs = dx.buf.double()
    tot = dxs.abs().sum(0)
    lim = 2e-6 if f32 else 2e-3
    assert float((dxs.sum(0).abs() / tot).max()) < lim and float(((dxs * xhat).sum(0).abs() / tot).max()) < lim


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('k,s,p,T,H,W,C_', [((3, 3, 3), (1, 1, 1), (1, 1, 1), 4, 14, 14, 256),
                                            ((1, 3, 3), (1, 2, 2), (0, 1, 1), 4, 56, 56, 64)])
def test_maxpool_bounds_and_gradient_mass_at_full_size(gpu, k, s, p, T, H, W, C_, dtype):
    tdt = ops.TORCH_DTYPE[dtype]
    g = torch.Generator().manual_seed(13)
    x = ops.new_act(N_CLIPS, T, H, W, C_, dtype, gpu)
    x.buf.copy_(torch.randn(x.buf.shape, generator=g).relu_().to(tdt))
    To, Ho, Wo = ops.conv_out_dims(x, k, s, p)
    y = ops.new_act(N_CLIPS, To, Ho, Wo, C_, dtype, gpu)
    idx = torch.zeros(y.rows, ops.cp8(C_), dtype=torch.uint8, device=gpu)
    d = ops.pool_desc(dtype, x, y, k, s, p)
    ops.call('dv_maxpool3d_fwd', d, x, y, idx)
    xs, ys = x.buf.view(N_CLIPS, T, H, W, C_), y.buf.view(N_CLIPS, To, Ho, Wo, C_)
    # the centre tap of every window is inside it: y >= x at the window centres; and y never exceeds the global maximum
    ctr = xs[:, (k[0] // 2 - p[0])::s[0], (k[1] // 2 - p[1])::s[1], (k[2] // 2 - p[2])::s[2]][:, :To, :Ho, :Wo]
    assert bool((ys >= ctr).all()) and float(ys.float().max()) == float(xs.float().max())
    assert int(idx.max()) < k[0] * k[1] * k[2]
    dy = y.like()
    dy.buf.copy_(torch.randn(dy.buf.shape, generator=g).to(tdt))
    dx = x.like()
    ops.call('dv_maxpool3d_bwd', d, dy, idx, dx, 0)
    torch.cuda.synchronize()
    sy, sx = dy.buf.double().sum(0), dx.buf.double().sum(0)               # per channel: every dy lands on exactly one input
    # (an input that wins several windows holds the fp32 / bf16 SUM of their gradients: one rounding per such element)
    assert float((sy - sx).abs().max()) <= (2e-7 if dtype == DV_F32 else 2e-3) * float(dy.buf.double().abs().sum(0).max())


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_full_size_step_is_batch_symmetric_and_loss_matches_its_logits(gpu, mode):
    """S3D-G SimCLR_Naked, 64 samples x 2 views of 8x112x112 (the bench workload), in fp32 (the headline arithmetic) and bf16"""
    from dualvar_amd import model as M
    torch.manual_seed(0)
    m = M.SimCLR_Naked('s3dg', 128, 0.07, False)
    m.set_compute_dtype(mode).train().to(gpu)
    g = torch.Generator().manual_seed(5)
    block = torch.randn(64, 2, 3, 8, 112, 112, generator=g).to(gpu)
    perm = torch.randperm(64, generator=g).to(gpu)
    with torch.no_grad():
        r1 = m(block)
        r2 = m(block[perm])
    lg = r1['clip_logits'].float().cpu()
    assert lg.shape == (128, 127) and bool(torch.isfinite(lg).all())
    # the loss is the cross-entropy of the returned logits (positive in column 0)
    ce = float(torch.nn.functional.cross_entropy(lg, torch.zeros(128, dtype=torch.long)))
    assert abs(ce - float(r1['clip_contrast_loss'])) < 1e-4
    # batch statistics do not depend on the order of the samples: same loss, and row i's positive logit moves with it
    # (a permutation regroups the partial sums of the statistics; the random-init S3D-G amplifies that rounding-sized change --
    # DESIGN section 2 -- which bf16 storage of every activation then multiplies)
    dl = abs(float(r1['clip_contrast_loss']) - float(r2['clip_contrast_loss']))
    pos1, pos2 = lg[:64, 0], r2['clip_logits'].float().cpu()[:64, 0]
    dp = float((pos1[perm.cpu()] - pos2).abs().max()) / float(pos1.abs().max())
    print(f'{mode}: |dloss| under a batch permutation {dl:.3e}, positive logits {dp:.3e} (relative)')
    assert dl < (2e-3 if mode == 'fp32' else 5e-2) and dp < (2e-2 if mode == 'fp32' else 0.15)


def test_full_size_fp32_training_step(gpu):
    """The headline bench step itself (S3D-G SimCLR_Naked fp32, 64 x 2 clips of 8x112x112): forward, backward, SGD.  Finite
    gradients everywhere, the step is reproducible bit for bit from the same state, and the loss goes down along the gradient for a
    small enough step (a wrong tile or a wrong split in any big-M weight gradient breaks the last two)."""
    from dualvar_amd import model as M
    from dualvar_amd.optim import SGD

    def run(lr, steps):
        torch.manual_seed(0)
        m = M.SimCLR_Naked('s3dg', 128, 0.07, False)
        m.set_compute_dtype('fp32').train().to(gpu)
        block = torch.randn(64, 2, 3, 8, 112, 112, generator=torch.Generator().manual_seed(5)).to(gpu)
        opt = SGD([p for p in m.parameters() if p.requires_grad], lr=lr, momentum=0.0, weight_decay=0.0, stores=m.stores())
        losses, grads = [], None
        for _ in range(steps):
            ret = m(block)
            opt.zero_grad()
            ret['clip_contrast_loss'].backward()
            if grads is None:
                grads = [st.grad.detach().clone() for st in m.stores()]
            opt.step()
            losses.append(float(ret['clip_contrast_loss']))
        return losses, grads

    _, g0 = run(0.0, 1)
    assert all(bool(torch.isfinite(g).all()) for g in g0)
    gn2 = float(sum(float(g.double().pow(2).sum()) for g in g0))
    assert gn2 > 0
    # first-order decrease: L(w - lr g) - L(w) = -lr |g|^2 (1 + O(lr)); lr chosen for a predicted decrease of 0.02 on a loss of ~4.8
    lr = 0.02 / gn2
    l1, g1 = run(lr, 2)
    l2, g2 = run(lr, 2)
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(g1, g2)), 'the fp32 step must be bit-reproducible'
    assert all(torch.equal(a, b) for a, b in zip(g0, g1))
    print(f'|g|^2 {gn2:.4e}, lr {lr:.3e}: loss {l1[0]:.6f} -> {l1[1]:.6f}; predicted decrease 0.02, observed {l1[0] - l1[1]:.4e}')
    assert 0.01 < l1[0] - l1[1] < 0.03


def test_full_size_r21d_tsv4_step(gpu):
    """BASELINE configs[2] at its per-GPU size: R(2+1)D SimCLR_TimeSeriesV4 (clip + shuffle-rank + tc heads; model/simclr.py:135-400,
    backbone/r21d.py:176-266), 32 samples x 3 views of 8x112x112 in fp32 -- three encoder passes per step.  Finite gradients, the
    step is bit-reproducible from the same state, and the summed loss goes down along the gradient by the first-order amount (a
    wrong tile, tap or split in any of the large-M kernels this size alone reaches -- 144 / 230 / 460 mid channels at 2.4 M / 602 k
    / 150 k rows -- breaks the last two)."""
    from dualvar_amd import model as M
    from dualvar_amd.optim import SGD
    import types

    def run(lr, steps):
        torch.manual_seed(0)
        np.random.seed(1234)
        m = M.SimCLR_TimeSeriesV4('r21d', 128, 0.07, False, args=types.SimpleNamespace(shufflerank_theta=0.05))
        m.set_compute_dtype('fp32').train().to(gpu)
        block = torch.randn(32, 3, 3, 8, 112, 112, generator=torch.Generator().manual_seed(5)).to(gpu)
        opt = SGD([p for p in m.parameters() if p.requires_grad], lr=lr, momentum=0.0, weight_decay=0.0, stores=m.stores())
        losses, grads = [], None
        for _ in range(steps):
            np.random.seed(77)                          # the same segment shuffles in every step: the loss is one function of the weights
            ret = m(block)
            loss = ret['clip_contrast_loss']
            for k_ in ret:
                if 'loss' in k_ and 'clip' not in k_:
                    loss = loss + ret[k_]
            opt.zero_grad()
            loss.backward()
            if grads is None:
                grads = [st.grad.detach().clone() for st in m.stores()]
            opt.step()
            losses.append(float(loss))
        return losses, grads

    _, g0 = run(0.0, 1)
    assert all(bool(torch.isfinite(g).all()) for g in g0)
    gn2 = float(sum(float(g.double().pow(2).sum()) for g in g0))
    assert gn2 > 0
    lr = 0.02 / gn2
    l1, g1 = run(lr, 2)
    l2, g2 = run(lr, 2)
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(g1, g2)), 'the fp32 step must be bit-reproducible'
    assert all(torch.equal(a, b) for a, b in zip(g0, g1))
    print(f'r21d tsv4: |g|^2 {gn2:.4e}, lr {lr:.3e}: loss {l1[0]:.6f} -> {l1[1]:.6f}; predicted decrease 0.02, observed {l1[0] - l1[1]:.4e}')
    assert 0.01 < l1[0] - l1[1] < 0.03


def test_full_size_16_frame_step_bf16(gpu):
    """BASELINE configs[1]: S3D-G SimCLR_Naked bf16, batch 64 (x 2 views), 16 x 112 x 112 clips -- one full training step at
    that size: the loss is the cross-entropy of the returned logits, every gradient is finite, and the batch symmetry of
    the forward holds."""
    from dualvar_amd import model as M
    from dualvar_amd.optim import SGD
    torch.manual_seed(0)
    m = M.SimCLR_Naked('s3dg', 128, 0.07, False)
    m.set_compute_dtype('bf16').train().to(gpu)
    g = torch.Generator().manual_seed(5)
    block = torch.randn(64, 2, 3, 16, 112, 112, generator=g).to(gpu)
    opt = SGD([p for p in m.parameters() if p.requires_grad], lr=0.003, momentum=0.9, weight_decay=1e-4, stores=m.stores())
    ret = m(block)
    lg = ret['clip_logits'].detach().float().cpu()
    assert lg.shape == (128, 127) and bool(torch.isfinite(lg).all())
    ce = float(torch.nn.functional.cross_entropy(lg, torch.zeros(128, dtype=torch.long)))
    assert abs(ce - float(ret['clip_contrast_loss'])) < 1e-4
    opt.zero_grad()
    ret['clip_contrast_loss'].backward()
    assert all(bool(torch.isfinite(st.grad).all()) for st in m.stores())
    gn = float(sum(float(st.grad.double().pow(2).sum()) for st in m.stores()) ** 0.5)
    assert gn > 0
    opt.step()
    perm = torch.randperm(64, generator=g).to(gpu)
    with torch.no_grad():
        r1, r2 = m(block), m(block[perm])
    # (bf16 storage after one SGD step on the random-init net: a permutation regroups the statistics' partial sums, and bf16
    # rounding of the activations amplifies that to a few 1e-2 on a loss of ~4.8 -- observed 0.02 .. 0.055 from run to run)
    assert abs(float(r1['clip_contrast_loss']) - float(r2['clip_contrast_loss'])) < 0.1


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_moco_naked_full_queue_step(gpu, dtype):
    """BASELINE configs[3]: S3D-G MoCo_Naked with the paper's queue, K = 65 536 (model/moco.py:28-239): one full training
    step -- logits [B, 1 + K] with the positive in column 0, the loss is their cross-entropy, the B keys land in the queue
    at the pointer, the pointer advances by B, the key encoder moves by the momentum rule and receives no gradient."""
    from dualvar_amd import model as M
    from dualvar_amd.optim import SGD
    torch.manual_seed(0)
    K, B = 65536, 16
    m = M.MoCo_Naked('s3dg', 128, K, 0.999, 0.07, False)
    m.set_compute_dtype(dtype).train().to(gpu)
    g = torch.Generator().manual_seed(9)
    block = torch.randn(B, 2, 3, 8, 112, 112, generator=g).to(gpu)
    opt = SGD([p for p in m.parameters() if p.requires_grad], lr=0.003, momentum=0.9, weight_decay=1e-4, stores=m.stores())
    q0 = m.queue.detach().clone()
    kq0 = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith('encoder_k.') and v.dtype.is_floating_point}
    qq0 = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith('encoder_q.') and v.dtype.is_floating_point}
    ret = m(block)
    lg = ret['clip_logits'].detach().float().cpu()
    assert lg.shape == (B, 1 + K) and bool(torch.isfinite(lg).all())
    ce = float(torch.nn.functional.cross_entropy(lg, torch.zeros(B, dtype=torch.long)))
    assert abs(ce - float(ret['clip_contrast_loss'])) < 2e-4 * max(1.0, ce)
    # negatives: logits[:, 1:] = q . queue / T  -> |logit| <= 1/T for unit vectors
    assert float(lg.abs().max()) <= 1.0 / 0.07 * 1.01
    assert int(m.queue_ptr) == B
    q1 = m.queue.detach()
    assert torch.equal(q1[:, B:], q0[:, B:])                          # only the B columns at the pointer changed
    cols = q1[:, :B].float()
    assert float((cols.norm(dim=0) - 1).abs().max()) < (1e-5 if dtype == 'fp32' else 1e-2)     # unit-norm keys
    # positive logit = q . k / T with k = the enqueued key: recover q . k from column 0 and compare with the range
    opt.zero_grad()
    ret['clip_contrast_loss'].backward()
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for n, p in m.named_parameters() if n.startswith('encoder_k.'))
    assert any(float(p.grad.abs().sum()) > 0 for n, p in m.named_parameters() if n.startswith('encoder_q.') and p.grad is not None)
    # momentum rule (moco.py:104-107), applied at the start of the forward: k <- 0.999 k + 0.001 q
    sd = m.state_dict()
    worst = 0.0
    for k, v0 in kq0.items():
        if 'running' in k or 'num_batches' in k:
            continue
        want = 0.999 * v0 + 0.001 * qq0['encoder_q.' + k[len('encoder_k.'):]]
        worst = max(worst, float((sd[k] - want).abs().max()))
    assert worst < 1e-6, worst
    opt.step()
