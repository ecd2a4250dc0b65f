"""Common machinery of the HIP backbones.

A backbone is an nn.Module whose sub-modules are *parameter holders* (nn.Conv3d / nn.BatchNorm3d /
nn.Linear instances carrying the reference's names, shapes and initialisation -- so state_dict()
round-trips with reference checkpoints) plus an `emit(plan, x)` method that lays the network out as
engine ops.  Compute never goes through those holders' own forward()."""
import os
import weakref

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..engine import Comm, Op, ParamStore, Plan
from ..ops import DV_BF16, DV_F32
from ..utils.transforms import FrameBatch

_DTYPES = {'bf16': DV_BF16, 'bfloat16': DV_BF16, 'fp32': DV_F32, 'float32': DV_F32, 'f32': DV_F32}


def default_dtype():
    """fp32 -- the reference's arithmetic (no autocast anywhere in it) and the mode whose results are pinned to it at 1e-3;
    bf16 storage is the opt-in throughput mode (model.set_compute_dtype('bf16'), pretrain.py --dtype bf16, DUALVAR_DTYPE=bf16)"""
    return _DTYPES[os.environ.get('DUALVAR_DTYPE', 'fp32').lower()]


class IngestOp(Op):
    """NCDHW fp32 clips (the reference's input layout) -> NDHWC compute dtype; optional Normalize and
    temporal segment shuffle fused in (pretrain.py:386-389, simclr.py:378-383)."""

    def __init__(self, plan, N, T, H, W, n_seg, pad=0):
        super().__init__(plan)
        # pad > 0: frames carry a zero border of `pad` pixels (written once, here) -- the stem conv's padding
        self.pad = pad
        self.y = plan.act(N, T, H + 2 * pad, W + 2 * pad, 3, cpitch=4, grad=False, zero=True)
        self.y.hw_pad = pad
        self.N, self.T, self.H, self.W, self.n_seg = N, T, H, W, n_seg
        self.src = self.perm = self.mean = self.istd = self.cmean = None
        self.stride_n = 3 * T * H * W

    def bind(self, x, perm, mean, istd):
        self.src, self.perm, self.mean, self.istd = x, perm, mean, istd
        if isinstance(x, FrameBatch):
            if self.cmean is None:
                self.cmean = torch.empty(self.N * self.T, dtype=torch.float32, device=x.device)
            if x.blur is not None and getattr(self, 'blur_tmp', None) is None:      # quantised frames in front of the blur
                self.blur_tmp = torch.empty(self.N * self.T * self.H * self.W * 3, dtype=torch.uint8, device=x.device)
            return
        self.stride_n = x.stride(0) if x.dim() == 5 else 3 * self.T * self.H * self.W

    def _launch(self, stream):
        p = self.plan
        if isinstance(self.src, FrameBatch):
            # decoded uint8 frames + augmentation table -> the same NDHWC stem input (utils/transforms.py: FrameBatch)
            fb = self.src
            L.check(p.lib.dv_augment_ingest(p.dtype, fb.frames.data_ptr(), fb.frames.shape[0], fb.frames.shape[1],
                                            fb.frames.shape[2], fb.table.data_ptr(), self.N, self.T, self.H, self.W,
                                            self.y.ptr, self.y.ld, self.pad,
                                            self.mean.data_ptr() if self.mean is not None else 0,
                                            self.istd.data_ptr() if self.istd is not None else 0,
                                            self.perm.data_ptr() if self.perm is not None else 0,
                                            self.n_seg if self.perm is not None else 0, self.cmean.data_ptr(),
                                            fb.blur.data_ptr() if fb.blur is not None else 0,
                                            self.blur_tmp.data_ptr() if fb.blur is not None else 0, stream),
                    'dv_augment_ingest')
            return
        L.check(p.lib.dv_ingest_ncdhw_pad(p.dtype, self.src.data_ptr(), self.y.ptr, self.N, 3, self.T, self.H, self.W,
                                          self.stride_n, self.y.ld,
                                          self.mean.data_ptr() if self.mean is not None else 0,
                                          self.istd.data_ptr() if self.istd is not None else 0,
                                          self.perm.data_ptr() if self.perm is not None else 0,
                                          self.n_seg if self.perm is not None else 0, self.pad, stream), 'dv_ingest_ncdhw_pad')

    def launches(self):
        return [_IngestStep(self)], []


class _IngestStep:
    """the source pointer changes per call, so this step binds its arguments when it runs"""
    __slots__ = ('name', 'kname', 'op', 'bytes', 'flops')

    def __init__(self, op):
        self.op, self.name, self.flops = op, 'ingest', 0
        self.kname = 'ingest<%s>' % ('f32' if op.plan.dtype == DV_F32 else 'bf16')
        self.bytes = op.N * op.T * op.H * op.W * (12 + 4 * ops.ESIZE[op.plan.dtype])

    def __call__(self, stream):
        self.op._launch(stream)


class _Token:
    pass


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, backbone, perm, want_map):
        plan = backbone._acquire_plan(x, perm is not None, want_map, with_grad=True)
        backbone._run_plan(plan, x, perm)
        plan.store.pending_backward += 1          # encoder passes of this step still to be back-propagated
        ctx.plan = plan
        ctx.token = _Token()
        plan._busy_ref = weakref.ref(ctx.token)
        ctx.want_map = want_map
        if want_map:
            return ops.act_to_ncdhw(plan.out_act)
        return plan.pooled.clone()

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        backbone_store = plan.store
        if getattr(backbone_store, '_sync_started', False):
            # the bucket-wise all-reduce of this arena (GradSync.attach) is already in flight: a further backward pass would add
            # local gradients to buckets that hold cross-rank sums, and the ranks would diverge silently.  The reference's loop
            # runs ONE loss.backward() per optimizer step (pretrain.py:447-451).
            raise RuntimeError('a backward pass wrote the gradient arena after its overlapped all-reduce had started; gradient '
                               'accumulation over several loss.backward() calls needs GradSync without attach()')
        backbone_store.attach_grads()
        # Bucket-wise gradient all-reduce from inside the backward list (parallel.GradSync.attach) only on the LAST encoder
        # backward of the step: the dual-head objectives run the encoder two or three times per step (simclr.py:354,385;
        # moco.py:492,551) and every pass accumulates into the same arena -- a bucket is final only once the last one
        # has written it.  (The heads' gradients of every pass are complete by then: a pass's head backward precedes its
        # encoder backward.)
        backbone_store.pending_backward = max(backbone_store.pending_backward - 1, 0)
        if backbone_store.pending_backward > 0:
            plan.grad_ready = None
        if ctx.want_map:
            a = plan.out_act.grad
            a.buf[:, a.off:a.off + a.C] = g.permute(0, 2, 3, 4, 1).reshape(-1, a.C).to(a.buf.dtype)
        else:
            plan.mean_op.dout.copy_(g)
        plan.run_backward()
        plan._busy_ref = None
        hook = plan.after_backward
        if hook is not None:
            hook(plan)
        return None, None, None, None, None


class HipBackbone(nn.Module):
    """Base class: owns (or is bound to) a ParamStore and a cache of launch plans."""
    feature_size = 0
    stem_pad = 3            # all four backbones start with a 7x7 / stride 2 / padding 3 conv on RGB (see Plan.conv)

    def __init__(self):
        super().__init__()
        self._store = None
        self._own_store = True
        self._plans = {}
        self._dtype = None
        self._fp8pw = False
        self._anchor = None
        self._norm = None
        self.comm = None
        self.after_backward = None      # set by the data-parallel wrapper (gradient all-reduce hook)
        self.grad_ready = None          # parallel.GradSync.attach: bucket-wise all-reduce from inside the backward list
        self.bucket_elems = 0

    # -- configuration
    def set_compute_dtype(self, name):
        """'fp32' | 'bf16' | 'fp8pw': bf16 storage with the 1x1x1 convs a network marks (the 2D3D-ResNet bottlenecks' conv1 /
        conv3, BASELINE configs[4]) on the fp8 matrix-core path -- forward and data gradient, per-tensor e4m3 / e5m2 scaling"""
        self._fp8pw = isinstance(name, str) and name.lower() in ('fp8pw', 'fp8')
        self._dtype = DV_BF16 if self._fp8pw else (_DTYPES[name] if isinstance(name, str) else name)
        self._plans.clear()
        return self

    def set_input_normalization(self, mean, std):
        """Fuse utils.transforms.Normalize (pretrain.py:280-282) into the ingest kernel."""
        if mean is None:
            self._norm = None
            return self
        self._norm = (torch.tensor(mean, dtype=torch.float32), 1.0 / torch.tensor(std, dtype=torch.float32))
        return self

    def bind_store(self, store):
        """Called by the owning objective so that backbone + heads share one arena."""
        self._store, self._own_store = store, False
        self.register_params(store)

    @property
    def store(self):
        if self._store is None:
            self._store = ParamStore()
            self.register_params(self._store)
        return self._store

    @property
    def dtype(self):
        if self._dtype is None:
            self._dtype = default_dtype()
        return self._dtype

    # -- to be provided by the concrete network
    def register_params(self, store):
        raise NotImplementedError

    def emit(self, plan, x):
        raise NotImplementedError

    # -- plans
    def prepare(self, device):
        L.require_device()
        st = self.store
        if not st.ready(device, self.dtype):
            st.materialize(device, self.dtype)
            self._plans.clear()
        st.refresh()
        if self._anchor is None or self._anchor.device != device:
            self._anchor = torch.zeros(1, device=device, requires_grad=True)
        if self._norm is not None and self._norm[0].device != device:
            self._norm = (self._norm[0].to(device), self._norm[1].to(device))

    def _acquire_plan(self, x, use_perm, want_map, with_grad):
        N, _, T, H, W = x.shape
        n_seg = 0
        key = (N, T, H, W, use_perm, want_map, with_grad, self.store.generation, self.dtype, self.training, self._fp8pw)
        lst = self._plans.setdefault(key, [])
        for pl in lst:
            ref = pl._busy_ref
            if ref is None or ref() is None:
                return pl
        comm = self.comm if self.comm is not None else Comm()
        pl = Plan(self.store, self.dtype, x.device, with_grad=with_grad, comm=comm, training=self.training)
        pl.fp8_pointwise = self._fp8pw
        pl.ingest = pl._push(IngestOp(pl, N, T, H, W, n_seg, pad=self.stem_pad))
        out = self.emit(pl, pl.ingest.y)
        pl.out_act = out
        pl.mean_op = None
        if not want_map:
            pl.pooled = pl.spatial_mean(out)
            pl.mean_op = pl.ops[-1]
        pl.finalize()
        pl._busy_ref = None
        pl.after_backward = None
        lst.append(pl)
        return pl

    def _run_plan(self, plan, x, perm):
        if isinstance(x, FrameBatch):
            pass
        elif x.dtype != torch.float32 or x.stride()[1:] != (x.shape[2] * x.shape[3] * x.shape[4], x.shape[3] * x.shape[4], x.shape[4], 1):
            x = x.float().contiguous()
        plan._keep = x
        mean, istd = self._norm if self._norm is not None else (None, None)
        plan.ingest.n_seg = perm.shape[1] if perm is not None else 0
        plan.ingest.bind(x, perm, mean, istd)
        plan.after_backward = self.after_backward
        plan.grad_ready = self.grad_ready
        if self.grad_ready is not None and not plan.bucket_starts:
            from ..parallel import bucket_ranges
            plan.bucket_starts = tuple(a for a, _ in bucket_ranges(self.store.total, self.bucket_elems))
        with torch.no_grad():
            if self.store._fp8_stale:            # a plan built after prepare() registered new fp8 weight copies
                self.store._refresh_fp8()
            if self.training:
                self.store.bump_bn_counters()
            plan.run_forward()

    def _call(self, x, perm, want_map):
        if x.dim() != 5 or x.shape[1] != 3:
            raise ValueError('expected clips [N, 3, T, H, W], got %s' % (tuple(x.shape),))
        if self.stem_pad and x.shape[4] % 2:
            raise ValueError('frame width must be even (pixel-pair stem layout), got %d' % x.shape[4])
        if not x.is_cuda:
            raise L.DualVarHipError('dualvar_amd backbones run on the MI355X only (input is on %s)' % x.device)
        self.prepare(x.device)
        if perm is not None:
            perm = torch.as_tensor(np.ascontiguousarray(perm), dtype=torch.int32).to(x.device) \
                if not isinstance(perm, torch.Tensor) else perm.to(device=x.device, dtype=torch.int32).contiguous()
        if not self.training:
            # module.eval() (classifier.py's test / retrieval passes, the 'last'-layer finetune): BatchNorm with the
            # running statistics, forward only -- the result carries no autograd history
            if torch.is_grad_enabled() and x.requires_grad:
                raise NotImplementedError('backward through an eval-mode backbone is not built')
        elif torch.is_grad_enabled() and any(s.tensor.requires_grad for s in self.store.slots):
            return _BackboneFn.apply(self._anchor, x, self, perm, want_map)
        plan = self._acquire_plan(x, perm is not None, want_map, with_grad=False)
        self._run_plan(plan, x, perm)
        return ops.act_to_ncdhw(plan.out_act) if want_map else plan.pooled.clone()

    def forward(self, x):
        """[N,3,T,H,W] -> [N,feature_size,T',H',W'] (post-ReLU), the reference module contract."""
        return self._call(x, None, True)

    def forward_pooled(self, x, perm=None):
        """[N,3,T,H,W] -> [N,feature_size] = AdaptiveAvgPool3d(1)(backbone(x)), pooled inside the plan.
        perm [N, n_series] applies the reference's temporal segment shuffle while ingesting."""
        return self._call(x, perm, False)


# ------------------------------------------------------------------ shared emit helpers
def conv_geometry(conv):
    return tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding)


def emit_conv_bn(plan, conv, bn, x, relu=True, residual=None, out=None, fp8=False):
    """conv (bias-free) -> train-mode BN (+residual) (+ReLU).  fp8: the layer may run on the fp8 pointwise path when the
    plan's compute mode is 'fp8pw' (Plan.conv checks that it is a 1x1x1 stride-1 conv with 16-aligned channel pitches)."""
    k, s, p = conv_geometry(conv)
    raw = plan.conv(plan.store.slot(conv.weight), x, k, s, p, fp8=fp8)
    return plan.bn(bn, raw, relu=relu, residual=residual, out=out)


def register_conv_bn(store, conv, bn, first=False):
    rgb = conv.in_channels == 3
    # the RGB stems (7x7, stride 2, padding 3) store their kernel rows 8 taps wide: see Plan.conv
    wide = 8 if (rgb and first and tuple(conv.kernel_size[1:]) == (7, 7)) else 0
    store.add_conv(conv.weight, cin_pitch=4 if rgb else None, need_dgrad=not first, kw_store=wide)
    if bn is not None:
        store.add_bn(bn)
