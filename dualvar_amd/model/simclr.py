"""SimCLR objectives on the HIP engine: SimCLR_Naked and the dual-head SimCLR_TimeSeriesV4
(reference model/simclr.py:19-127,130-400).  Constructor signatures, sub-module names (encoder_q.N,
series_proj_head.N), the returned dict keys and the logits layouts are the reference's; the compute is the
backbone launch plan + fused head / loss kernels.  Extra `*_rank0` entries carry the positive's rank for
the accuracy meters (they contain neither 'loss' nor 'logits', so the reference train loop ignores them)."""
import types

import numpy as np
import torch
import torch.nn as nn

from .. import functional as DF
from ..backbone.select_backbone import select_backbone
from ..engine import Comm, ParamStore
from ..utils.utils import gather_features


def _proj_modules(cin, cout):
    return [nn.Conv3d(cin, cin, kernel_size=1, bias=True), nn.ReLU(), nn.Conv3d(cin, cout, kernel_size=1, bias=True)]


class _Objective(nn.Module):
    """shared plumbing: one ParamStore per encoder, SyncBN communicator, input-normalisation passthrough"""

    def _bind(self, backbone, heads):
        store = ParamStore()
        backbone.bind_store(store)
        objs = []
        for a, b in heads:
            DF.ProjectionHead.register(store, a, b)
            objs.append(DF.ProjectionHead(store, a, b))
        return store, objs

    def stores(self):
        return [self.store]

    def set_compute_dtype(self, name):
        for m in self.modules():
            if hasattr(m, '_plans') and m is not self:
                m.set_compute_dtype(name)
        return self

    def set_input_normalization(self, mean, std):
        for m in self.modules():
            if hasattr(m, '_plans'):
                m.set_input_normalization(mean, std)
        return self

    def _sync_comm(self, *backbones):
        """SyncBatchNorm communicator (pretrain.py:244): created at the first forward, once the process
        group exists."""
        if self.distributed:
            for bb in backbones:
                if bb.comm is None:
                    bb.comm = Comm()


class _SimCLRLosses:
    def calc_clip_contrast_loss(self, features, n_views=2, prefix='clip_'):
        """simclr.py:56-99 / :183-229.  features [B, 2, dim] (L2-normalised).  Every rank evaluates the full
        [2N, 2N] NT-Xent over the gathered features, as the reference does."""
        B, nv, dim = features.shape
        assert nv == n_views == 2, features.shape
        allf = gather_features(features.contiguous(), self.distributed)
        N = allf.size(0)
        vm = allf.permute(1, 0, 2).reshape(nv * N, dim)                      # view-major rows
        loss, logits, rank0 = DF.ntxent(vm, vm, N, N, 0, self.T)
        labels = torch.zeros(logits.size(0), dtype=torch.long, device=logits.device)
        return {f'{prefix}logits': logits, f'{prefix}labels': labels, f'{prefix}contrast_loss': loss,
                f'{prefix}rank0': rank0}

    calc_contrast_loss = calc_clip_contrast_loss          # the reference calls it by this name


class SimCLR_Naked(_Objective, _SimCLRLosses):
    def __init__(self, network='s3d', dim=128, T=0.07, distributed=True, nonlinear=True):
        super().__init__()
        self.dim, self.T, self.distributed, self.nonlinear = dim, T, distributed, nonlinear
        backbone, self.param = select_backbone(network)
        fs = self.param['feature_size']
        self.encoder_q = nn.ModuleList([backbone, nn.AdaptiveAvgPool3d((1, 1, 1))])
        if nonlinear:
            self.encoder_q.extend(_proj_modules(fs, dim))
        self.criterion = nn.CrossEntropyLoss()
        heads = [(self.encoder_q[2], self.encoder_q[4])] if nonlinear else []
        self.store, hs = self._bind(backbone, heads)
        self._head = hs[0] if nonlinear else None

    def forward(self, block):
        B, nv = block.shape[:2]
        assert nv == 2
        self._sync_comm(self.encoder_q[0])
        pooled = self.encoder_q[0].forward_pooled(block.reshape(-1, *block.shape[2:]))
        z = self._head(pooled) if self._head is not None else pooled
        feats = DF.l2_normalize(z).view(B, nv, self.dim)
        return self.calc_clip_contrast_loss(feats, nv, 'clip_')


class SimCLR_TimeSeriesV4(_Objective, _SimCLRLosses):
    def __init__(self, network='s3d', dim=128, T=0.07, distributed=True, nonlinear=True, n_series=2, series_dim=64,
                 series_T=0.07, aligned_T=0.07, mode='clip-sr-tc', args=None):
        super().__init__()
        self.cnt = 0
        self.args = args if args is not None else types.SimpleNamespace(shufflerank_theta=0.05)
        self.dim, self.T, self.distributed, self.nonlinear = dim, T, distributed, nonlinear
        self.n_series, self.series_dim, self.series_T, self.aligned_T, self.mode = n_series, series_dim, series_T, aligned_T, mode
        self.with_clip, self.with_sr, self.with_tc = 'clip' in mode, 'sr' in mode, 'tc' in mode
        backbone, self.param = select_backbone(network)
        fs = self.param['feature_size']
        self.encoder_q = nn.ModuleList([backbone, nn.AdaptiveAvgPool3d((1, 1, 1))])
        has_clip_head = nonlinear and self.with_clip
        if has_clip_head:
            self.encoder_q.extend(_proj_modules(fs, dim))
        self.criterion = nn.CrossEntropyLoss()
        self.series_proj_head = nn.Sequential(*_proj_modules(fs, series_dim * n_series))
        heads = ([(self.encoder_q[2], self.encoder_q[4])] if has_clip_head else []) + \
            [(self.series_proj_head[0], self.series_proj_head[2])]
        self.store, hs = self._bind(backbone, heads)
        self._clip_head = hs[0] if has_clip_head else None
        self._series_head = hs[-1]

    def calc_ranking_loss(self, features, n_views=2, prefix='ranking_', weight=1.):
        """simclr.py:231-278: features [Bn, n_series, 2, series_dim]."""
        Bn, s, nv, dim = features.shape
        assert s == self.n_series and nv == n_views == 2 and dim == self.series_dim, features.shape
        vm = features.permute(0, 2, 1, 3).reshape(Bn, nv * s, dim)
        loss, logits = DF.rank_margin(vm, s, self.args.shufflerank_theta, 5.0, weight)
        labels = torch.zeros(logits.size(0), dtype=torch.long, device=logits.device)
        return {f'{prefix}margin_logits': logits, f'{prefix}margin_labels': labels, f'{prefix}margin_contrast_loss': loss}

    def calc_tc_contrast_loss(self, features, prefix='tc_'):
        """simclr.py:280-337: rows are this rank's 2B entries, columns all 2N; the (s x s)-averaged similarity
        equals the dot product of the series-mean vectors, which is what the kernel contracts."""
        B, nv, s, sd = features.shape
        assert s == self.n_series and sd == self.series_dim and nv == 2
        rank, world = 0, 1
        allf = gather_features(features.contiguous(), self.distributed)
        if self.distributed:
            import torch.distributed as dist
            rank, world = dist.get_rank(), dist.get_world_size()
        N = allf.size(0)
        n = N // world
        base = n * rank
        means = DF.group_mean(allf.permute(1, 0, 2, 3).reshape(nv * N, s, sd))      # [2N, sd] view-major
        rows = means.view(nv, N, sd)[:, base:base + n].reshape(nv * n, sd)
        loss, logits, rank0 = DF.ntxent(rows, means, n, N, base, self.aligned_T)
        labels = torch.zeros(logits.size(0), dtype=torch.long, device=logits.device)
        return {f'{prefix}logits': logits, f'{prefix}labels': labels, f'{prefix}contrast_loss': loss,
                f'{prefix}rank0': rank0}

    def forward(self, block):
        block = block.contiguous()
        B, NV, C, T, H, W = block.shape
        assert NV == 3
        s, sd = self.n_series, self.series_dim
        bb = self.encoder_q[0]
        self._sync_comm(bb)
        pooled = bb.forward_pooled(block.view(-1, C, T, H, W))                    # [3B, F]
        ret = {}
        if self.with_clip:
            z = self._clip_head(pooled) if self._clip_head is not None else pooled
            feats = DF.l2_normalize(z).view(B, NV, self.dim)[:, :2]
            ret.update(self.calc_clip_contrast_loss(feats, 2))
        series = DF.l2_normalize(self._series_head(pooled).view(B, NV, s, sd))
        if self.with_tc:
            ret.update(self.calc_tc_contrast_loss(series[:, :2]))
        if self.with_sr:
            perm = np.array([np.random.permutation(s) for _ in range(B)])           # simclr.py:378-381 (global numpy RNG)
            sp = bb.forward_pooled(block[:, 2], perm=perm)                           # shuffle fused into the ingest
            sf = self._series_head(sp).view(B, s, sd)
            idx = torch.as_tensor(perm, dtype=torch.long, device=sf.device).view(B, s, 1).expand_as(sf)
            sf = DF.l2_normalize(torch.scatter(sf, 1, idx, sf))                      # un-permute (:389-393)
            ret.update(self.calc_ranking_loss(torch.stack([series[:, 0], sf], dim=2), 2, 'aug_ranking_', weight=0.5))
            ret.update(self.calc_ranking_loss(torch.stack([series[:, 2], sf], dim=2), 2, 'unaug_ranking_', weight=0.5))
        return ret
