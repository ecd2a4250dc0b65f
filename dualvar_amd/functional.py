"""torch.autograd.Function wrappers over the fp32 head / loss kernels.

These are the small ops after the backbone (a few KB..MB per step): projection heads, L2
normalisation, the fused similarity + cross-entropy objectives.  All arithmetic is in the HIP
library; torch only provides autograd plumbing, views and allocation.  Parameter gradients are
accumulated straight into the owning ParamStore's gradient arena (like the backbone's)."""
import ctypes as C

import torch

from . import _lib as L
from . import ops
from .ops import DV_BIAS, DV_F32, DV_RELU, Act


def _lib():
    return L.load()


def _chk(rc, what):
    L.check(rc, what)


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _need_gpu(t):
    if not t.is_cuda:
        raise L.DualVarHipError('this op runs on the MI355X only; got a %s tensor (no CPU fallback)' % t.device)
    L.require_device()


class ProjectionHead:
    """pooled [n,F] -> Conv1x1x1(F->F)+bias -> ReLU -> Conv1x1x1(F->D)+bias   (simclr.py:45-50,167-180;
    moco.py:56-72,282-308).  `conv_a`, `conv_b` are nn.Conv3d parameter holders registered in `store`."""

    def __init__(self, store, conv_a, conv_b):
        self.store, self.a, self.b = store, conv_a, conv_b

    @staticmethod
    def register(store, conv_a, conv_b):
        for c in (conv_a, conv_b):
            store.add_conv(c.weight, need_dgrad=False)
            store.add_vec(c.bias)

    def __call__(self, pooled):
        return _HeadFn.apply(pooled, self)

    def _linear(self, x, conv, flags):
        st, lib = self.store, _lib()
        ws, bs = st.slot(conv.weight), st.slot(conv.bias)
        n, fin, fout = x.shape[0], ws.Cin, ws.Cout
        y = torch.empty(n, fout, dtype=torch.float32, device=x.device)
        assert fin == ws.cin_pitch and fout % 8 == 0, 'head widths must be multiples of 8'
        _linear_fwd(x, st.w_master(ws), ws.cin_pitch, st.w_master(bs), y, n, fin, fout, fout, bool(flags & DV_RELU))
        return y

    def _linear_bwd(self, dy, x, conv, need_dx=True):
        """accumulates dW, db into the gradient arena; returns dx"""
        st, lib, s = self.store, _lib(), ops.stream_ptr()
        ws, bs = st.slot(conv.weight), st.slot(conv.bias)
        n, fin, fout = x.shape[0], ws.Cin, ws.Cout
        if conv.weight.requires_grad:
            _chk(lib.dv_gemm_f32(fout, fin, n, dy.data_ptr(), 1, fout, x.data_ptr(), fin, 1, st.w_grad(ws), ws.cin_pitch,
                                 1.0, 1, s), 'head dW')
            _chk(lib.dv_colsum_f32(dy.data_ptr(), fout, n, fout, st.w_grad(bs), s), 'head db')
        if not need_dx:
            return None
        dx = torch.empty(n, fin, dtype=torch.float32, device=x.device)
        _chk(lib.dv_gemm_f32(n, fin, fout, dy.data_ptr(), fout, 1, st.w_master(ws), ws.cin_pitch, 1, dx.data_ptr(), fin,
                             1.0, 0, s), 'head dx')
        return dx


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pooled, head):
        _need_gpu(pooled)
        x = _f32c(pooled)
        h = head._linear(x, head.a, DV_RELU)
        z = head._linear(h, head.b, 0)
        ctx.head = head
        ctx.save_for_backward(x, h)
        return z

    @staticmethod
    def backward(ctx, dz):
        head = ctx.head
        x, h = ctx.saved_tensors
        head.store.attach_grads()
        dz = _f32c(dz)
        dh = head._linear_bwd(dz, h, head.b)
        dh2 = torch.empty_like(dh)
        _chk(_lib().dv_relu_bwd_f32(dh.data_ptr(), h.data_ptr(), dh.numel(), dh2.data_ptr(), ops.stream_ptr()), 'relu bwd')
        dx = head._linear_bwd(dh2, x, head.a, need_dx=ctx.needs_input_grad[0])
        return dx, None


class _L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1, eps=1e-12) on [..., D]"""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        xc = _f32c(x)
        D = xc.shape[-1]
        R = xc.numel() // D
        y = torch.empty_like(xc)
        nrm = torch.empty(R, dtype=torch.float32, device=x.device)
        _chk(_lib().dv_l2norm_fwd(xc.data_ptr(), R, D, 1e-12, y.data_ptr(), nrm.data_ptr(), ops.stream_ptr()), 'l2norm')
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        dy = _f32c(dy)
        D = y.shape[-1]
        dx = torch.empty_like(y)
        _chk(_lib().dv_l2norm_bwd(dy.data_ptr(), y.data_ptr(), nrm.data_ptr(), y.numel() // D, D, dx.data_ptr(),
                                  ops.stream_ptr()), 'l2norm bwd')
        return dx


def l2_normalize(x):
    return _L2NormFn.apply(x)


class _GroupMeanFn(torch.autograd.Function):
    """[R, G, D] -> [R, D] mean over G (the series-mean vector of the tc head)."""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        xc = _f32c(x)
        R, G, D = xc.shape
        y = torch.empty(R, D, dtype=torch.float32, device=x.device)
        _chk(_lib().dv_group_mean_f32(xc.data_ptr(), R, G, D, y.data_ptr(), ops.stream_ptr()), 'group mean')
        ctx.shape = (R, G, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        R, G, D = ctx.shape
        dy = _f32c(dy)
        dx = torch.empty(R, G, D, dtype=torch.float32, device=dy.device)
        _chk(_lib().dv_group_mean_bwd_f32(dy.data_ptr(), R, G, D, dx.data_ptr(), ops.stream_ptr()), 'group mean bwd')
        return dx


def group_mean(x):
    return _GroupMeanFn.apply(x)


class _GroupTileFn(torch.autograd.Function):
    """[R, D] -> [R, G, D] with every copy divided by G (adjoint of group_mean)."""

    @staticmethod
    def forward(ctx, x, G):
        _need_gpu(x)
        xc = _f32c(x)
        R, D = xc.shape
        y = torch.empty(R, G, D, dtype=torch.float32, device=x.device)
        _chk(_lib().dv_group_mean_bwd_f32(xc.data_ptr(), R, G, D, y.data_ptr(), ops.stream_ptr()), 'group tile')
        ctx.shape = (R, G, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        R, G, D = ctx.shape
        dy = _f32c(dy)
        dx = torch.empty(R, D, dtype=torch.float32, device=dy.device)
        _chk(_lib().dv_group_mean_f32(dy.data_ptr(), R, G, D, dx.data_ptr(), ops.stream_ptr()), 'group tile bwd')
        return dx, None


def group_tile_div(x, G):
    return _GroupTileFn.apply(x, G)


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, cols, n_local, N, row_index0, inv_T):
        _need_gpu(rows)
        rows, cols = _f32c(rows), _f32c(cols)
        R, D = rows.shape
        dev = rows.device
        logits = torch.empty(R, 2 * N - 1, dtype=torch.float32, device=dev)
        loss_rows = torch.empty(R, dtype=torch.float32, device=dev)
        rank0 = torch.empty(R, dtype=torch.int32, device=dev)
        dsim = torch.empty(R, 2 * N, dtype=torch.float32, device=dev)
        lib, s = _lib(), ops.stream_ptr()
        _chk(lib.dv_ntxent_fwd(rows.data_ptr(), cols.data_ptr(), R, n_local, N, D, row_index0, inv_T, logits.data_ptr(),
                               loss_rows.data_ptr(), rank0.data_ptr(), dsim.data_ptr(), s), 'dv_ntxent_fwd')
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _chk(lib.dv_mean_f32(loss_rows.data_ptr(), R, loss.data_ptr(), s), 'mean')
        ctx.save_for_backward(rows, cols, dsim)
        ctx.mark_non_differentiable(logits, rank0)
        return loss, logits, rank0

    @staticmethod
    def backward(ctx, gloss, _gl, _gr):
        rows, cols, dsim = ctx.saved_tensors
        R, D = rows.shape
        C2 = cols.shape[0]
        lib, s = _lib(), ops.stream_ptr()
        drows = torch.empty_like(rows)
        dcols = torch.empty_like(cols)
        _chk(lib.dv_gemm_f32(R, D, C2, dsim.data_ptr(), C2, 1, cols.data_ptr(), D, 1, drows.data_ptr(), D, 1.0, 0, s), 'ntxent drows')
        _chk(lib.dv_gemm_f32(C2, D, R, dsim.data_ptr(), 1, C2, rows.data_ptr(), D, 1, dcols.data_ptr(), D, 1.0, 0, s), 'ntxent dcols')
        # upstream gradient of the scalar loss (1.0 in the reference's plain sum of heads)
        return drows * gloss, dcols * gloss, None, None, None, None


def ntxent(rows, cols, n_local, N, row_index0, temperature):
    """NT-Xent of `rows` (this rank's view-major [2*n_local, D]) against `cols` (gathered view-major
    [2N, D]).  Returns (loss, logits [R, 2N-1] in the reference's [positive, negatives] layout, rank0)."""
    return _NTXentFn.apply(rows, cols, n_local, N, row_index0, 1.0 / temperature)


_INFONCE_WS = {}


class _InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, queue, inv_T):
        _need_gpu(q)
        q, k, queue = _f32c(q), _f32c(k), _f32c(queue)
        B, D = q.shape
        K = queue.shape[1]
        dev = q.device
        logits = torch.empty(B, K + 1, dtype=torch.float32, device=dev)
        dlog = torch.empty(B, K + 1, dtype=torch.float32, device=dev)
        loss_rows = torch.empty(B, dtype=torch.float32, device=dev)
        rank0 = torch.empty(B, dtype=torch.int32, device=dev)
        dq = torch.empty(B, D, dtype=torch.float32, device=dev)
        lib, s = _lib(), ops.stream_ptr()
        wsb = int(lib.dv_infonce_workspace(B, D, K))
        # ordered K-split of dq: zeroed once, the kernel leaves its tickets zero.  One workspace per STREAM: two launches that
        # share tickets from different streams would count each other's arrivals.
        key = (dev, wsb, s)
        ws = _INFONCE_WS.get(key)
        if ws is None and wsb:
            from ._lib import register_ticket_workspace
            ws = _INFONCE_WS[key] = register_ticket_workspace(torch.zeros(wsb // 4, dtype=torch.float32, device=dev))
        _chk(lib.dv_infonce_fwd(q.data_ptr(), k.data_ptr(), queue.data_ptr(), B, D, K, inv_T, logits.data_ptr(),
                                loss_rows.data_ptr(), rank0.data_ptr(), dlog.data_ptr(), dq.data_ptr(),
                                ws.data_ptr() if ws is not None else 0, wsb, s), 'dv_infonce_fwd')
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _chk(lib.dv_mean_f32(loss_rows.data_ptr(), B, loss.data_ptr(), s), 'mean')
        ctx.save_for_backward(dq)
        ctx.mark_non_differentiable(logits, rank0)
        return loss, logits, rank0

    @staticmethod
    def backward(ctx, gloss, _gl, _gr):
        (dq,) = ctx.saved_tensors
        return dq * gloss, None, None, None


def infonce(q, k, queue, temperature):
    """MoCo InfoNCE: logits [B, 1+K] = [q.k, q.queue]/T (queue [D, K], no gradient to k / queue)."""
    return _InfoNCEFn.apply(q, k, queue, 1.0 / temperature)


class _RankMarginFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, s, theta, clip, weight):
        _need_gpu(feats)
        f = _f32c(feats)
        Bn, n2, D = f.shape
        dev = f.device
        logits = torch.empty(Bn * n2, n2 - 1, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        df = torch.empty_like(f)
        scr = torch.empty(Bn, dtype=torch.float32, device=dev)
        _chk(_lib().dv_rank_margin(f.data_ptr(), Bn, s, D, theta, clip, weight, logits.data_ptr(), loss.data_ptr(),
                                   df.data_ptr(), scr.data_ptr(), ops.stream_ptr()), 'dv_rank_margin')
        ctx.save_for_backward(df)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _gl):
        (df,) = ctx.saved_tensors
        return df * gloss, None, None, None, None


def rank_margin(feats_vm, n_series, theta, clip, weight):
    """feats_vm [Bn, 2*n_series, D] (view-major per sample).  clip <= 0 disables the clamp (MoCo variant)."""
    return _RankMarginFn.apply(feats_vm, n_series, float(theta), float(clip) if clip else 0.0, float(weight))


# ---------------------------------------------------------------------------------------------------------------
# downstream classifier head (model/classifier.py:10-70, trained by classifier.py:422-498)
def _linear_fwd(x, w_ptr, w_pitch, b_ptr, y, n, fin, fout, ldy, relu):
    """y[n][:fout] = act(x[n][:fin] @ W^T + b), W stored [fout][w_pitch] in the arena: one fp32 MFMA GEMM (a workgroup per 32x32
    tile, K over its four waves) with bias and ReLU in the epilogue"""
    d = L.GemmDesc()
    d.A, d.B, d.C, d.bias = x.data_ptr(), w_ptr, y.data_ptr(), b_ptr
    d.sam, d.sak, d.sbk, d.sbn, d.ldc = fin, 1, 1, w_pitch, ldy
    d.M, d.N, d.K, d.flags, d.alpha = n, fout, fin, (DV_RELU if relu else 0), 1.0
    _chk(_lib().dv_gemm_f32_ex(C.byref(d), ops.stream_ptr()), 'linear fwd')


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b (+ReLU) with W, b in a ParamStore arena; dW, db accumulate into the gradient arena"""

    @staticmethod
    def forward(ctx, x, anchor, store, lin, relu):
        _need_gpu(x)
        xc = _f32c(x)
        ws, bs = store.slot(lin.weight), store.slot(lin.bias)
        n, fin, fout = xc.shape[0], ws.Cin, ws.Cout
        assert fin == ws.cin_pitch, 'linear input widths must be multiples of 8'
        fp = (fout + 7) // 8 * 8
        y = torch.empty(n, fp, dtype=torch.float32, device=x.device)
        if fp != fout:
            y[:, fout:].zero_()
        _linear_fwd(xc, store.w_master(ws), ws.cin_pitch, store.w_master(bs), y, n, fin, fout, fp, relu)
        ctx.store, ctx.lin, ctx.relu = store, lin, relu
        ctx.save_for_backward(xc, y)
        return y[:, :fout]

    @staticmethod
    def backward(ctx, dy):
        st, lin, lib, s = ctx.store, ctx.lin, _lib(), ops.stream_ptr()
        x, y = ctx.saved_tensors
        st.attach_grads()
        ws, bs = st.slot(lin.weight), st.slot(lin.bias)
        n, fin, fout = x.shape[0], ws.Cin, ws.Cout
        g = _f32c(dy)
        if ctx.relu:
            g2 = torch.empty_like(g)
            yv = y[:, :fout].contiguous()
            _chk(lib.dv_relu_bwd_f32(g.data_ptr(), yv.data_ptr(), g.numel(), g2.data_ptr(), s), 'linear relu bwd')
            g = g2
        if lin.weight.requires_grad:
            _chk(lib.dv_gemm_f32(fout, fin, n, g.data_ptr(), 1, fout, x.data_ptr(), fin, 1, st.w_grad(ws), ws.cin_pitch, 1.0, 1, s),
                 'linear dW')
            _chk(lib.dv_colsum_f32(g.data_ptr(), fout, n, fout, st.w_grad(bs), s), 'linear db')
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(n, fin, dtype=torch.float32, device=x.device)
            _chk(lib.dv_gemm_f32(n, fin, fout, g.data_ptr(), fout, 1, st.w_master(ws), ws.cin_pitch, 1, dx.data_ptr(), fin, 1.0, 0, s),
                 'linear dx')
        return dx, None, None, None, None


def _anchor_for(x, *params):
    """The parameters of these ops live in an arena, not in the autograd graph: when the INPUT carries no gradient (a
    frozen backbone) a dummy requires-grad input keeps the op on the tape so that its backward still fills the arena."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return torch.zeros((), device=x.device, requires_grad=True)
    return None


def linear(x, store, lin, relu=False):
    return _LinearFn.apply(x, _anchor_for(x, lin.weight, lin.bias), store, lin, relu)


class _BN1dTrainFn(torch.autograd.Function):
    """train-mode nn.BatchNorm1d on [B, F] fp32 (model/classifier.py:29-32): batch statistics, running-statistic update,
    backward through the same dv_bn_* kernels as the 3-D layers (one 'tile' = the batch)"""

    @staticmethod
    def forward(ctx, x, anchor, store, bn):
        _need_gpu(x)
        xc = _f32c(x)
        lib, s = _lib(), ops.stream_ptr()
        n, Fd = xc.shape
        assert Fd % 8 == 0
        gs, bs = store.slot(bn.weight), store.slot(bn.bias)
        dev = x.device
        part = torch.empty(2 * Fd, dtype=torch.float32, device=dev)
        local = torch.empty(2 * Fd + 1, dtype=torch.float32, device=dev)
        mean, invstd, scale, shift = (torch.empty(Fd, dtype=torch.float32, device=dev) for _ in range(4))
        _chk(lib.dv_bn_rows_partials_f32(xc.data_ptr(), Fd, n, Fd, part.data_ptr(), s), 'bn1d partials')
        mom = float(bn.momentum if bn.momentum is not None else 0.1)
        _chk(lib.dv_bn_stats_finalize(part.data_ptr(), 1, n, Fd, n, Fd, local.data_ptr(), store.w_master(gs), store.w_master(bs),
                                      float(bn.eps), mom, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), mean.data_ptr(),
                                      invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), s), 'bn1d finalize')
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        y = torch.empty_like(xc)
        _chk(lib.dv_bn_apply(DV_F32, xc.data_ptr(), Fd, scale.data_ptr(), shift.data_ptr(), 0, 0, y.data_ptr(), Fd, n, Fd, 0, s),
             'bn1d apply')
        ctx.store, ctx.bn = store, bn
        ctx.save_for_backward(xc, y, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        st, bn, lib, s = ctx.store, ctx.bn, _lib(), ops.stream_ptr()
        x, y, mean, invstd = ctx.saved_tensors
        st.attach_grads()
        n, Fd = x.shape
        g = _f32c(dy)
        gs, bs = st.slot(bn.weight), st.slot(bn.bias)
        sums = torch.zeros(2 * Fd, dtype=torch.float32, device=x.device)
        ws = torch.zeros(int(lib.dv_bn_bwd_reduce_workspace(n, Fd)) // 4, dtype=torch.float32, device=x.device)   # (fresh per call)
        no_mask = 32                                                     # DV_NO_RELU_MASK
        _chk(lib.dv_bn_bwd_reduce(DV_F32, g.data_ptr(), Fd, y.data_ptr(), Fd, x.data_ptr(), Fd, mean.data_ptr(), invstd.data_ptr(),
                                  n, Fd, no_mask, sums.data_ptr(), 1, ws.data_ptr(), s), 'bn1d bwd reduce')
        dx = torch.empty_like(x)
        _chk(lib.dv_bn_bwd_apply(DV_F32, g.data_ptr(), Fd, y.data_ptr(), Fd, x.data_ptr(), Fd, mean.data_ptr(), invstd.data_ptr(),
                                 st.w_master(gs), sums.data_ptr(), 1, 1.0 / n, 1.0, st.w_grad(gs), st.w_grad(bs), dx.data_ptr(), Fd,
                                 0, 0, n, Fd, no_mask, s), 'bn1d bwd apply')
        return dx, None, None, None


def batchnorm1d_train(x, store, bn):
    return _BN1dTrainFn.apply(x, _anchor_for(x, bn.weight, bn.bias), store, bn)


class _RowScaleFn(torch.autograd.Function):
    """y = x * m elementwise on [B, F] fp32 (inverted-dropout mask); dx = dy * m"""

    @staticmethod
    def forward(ctx, x, m):
        _need_gpu(x)
        xc = _f32c(x)
        n, Fd = xc.shape
        y = torch.empty_like(xc)
        _chk(_lib().dv_gate_scale(DV_F32, xc.data_ptr(), Fd, m.data_ptr(), n, 1, Fd, y.data_ptr(), Fd, ops.stream_ptr()), 'dropout')
        ctx.save_for_backward(m)
        return y

    @staticmethod
    def backward(ctx, dy):
        (m,) = ctx.saved_tensors
        g = _f32c(dy)
        n, Fd = g.shape
        dx = torch.empty_like(g)
        _chk(_lib().dv_gate_scale(DV_F32, g.data_ptr(), Fd, m.data_ptr(), n, 1, Fd, dx.data_ptr(), Fd, ops.stream_ptr()), 'dropout bwd')
        return dx, None


def dropout(x, p):
    """nn.Dropout in training mode: the Bernoulli mask comes from torch's generator (it cannot reproduce the CPU
    reference's stream anyway), scaled by 1/(1-p); the multiply and its backward are HIP"""
    if p <= 0.0:
        return x
    mask = torch.empty(x.shape, dtype=torch.float32, device=x.device).bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
    return _RowScaleFn.apply(x, mask)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        _need_gpu(logits)
        lg = _f32c(logits)
        R, K = lg.shape
        tgt = target.to(device=lg.device, dtype=torch.int32).contiguous()
        rows = torch.empty(R, dtype=torch.float32, device=lg.device)
        dl = torch.empty_like(lg)
        rank0 = torch.empty(R, dtype=torch.int32, device=lg.device)
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        s = ops.stream_ptr()
        _chk(_lib().dv_softmax_ce_fwd(lg.data_ptr(), K, R, K, tgt.data_ptr(), rows.data_ptr(), dl.data_ptr(), K, rank0.data_ptr(), s),
             'softmax ce')
        _chk(_lib().dv_mean_f32(rows.data_ptr(), R, loss.data_ptr(), s), 'ce mean')
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(rank0)
        return loss, rank0

    @staticmethod
    def backward(ctx, dloss, _drank):
        (dl,) = ctx.saved_tensors
        return dl * dloss, None


def cross_entropy(logits, target):
    """nn.CrossEntropyLoss()(logits, target) -> (loss, rank of the target per row)"""
    return _CrossEntropyFn.apply(logits, target)
