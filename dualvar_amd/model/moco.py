"""MoCo objectives on the HIP engine: MoCo_Naked and the dual-head MoCo_TimeSeriesV4
(reference model/moco.py:28-239,242-573).

Differences from a line-by-line port, all result-preserving:
  * the momentum update is ONE fused kernel over the flat parameter arenas (k = m*k + (1-m)*q,
    moco.py:104-107,329-334) instead of a per-tensor loop;
  * the DDP batch shuffle (moco.py:129-173) moves no clips: BatchNorm statistics are already global
    (SyncBN, pretrain.py:244), so shuffling which rank encodes which key cannot change the keys; the
    torch.randperm draw is kept so the host RNG stream matches the reference;
  * InfoNCE reads the queue in place ([dim, K], no clone) and the tc head contracts series-mean vectors.
"""
import types

import numpy as np
import torch
import torch.nn as nn

from .. import functional as DF
from .. import ops
from ..backbone.select_backbone import select_backbone
from ..engine import ParamStore
from ..utils.transforms import FrameBatch
from ..utils.utils import concat_all_gather
from .simclr import _Objective, _proj_modules


class _MoCoBase(_Objective):
    def stores(self):
        return [self.store]                      # only the query side is trained / all-reduced

    def _make_encoder(self, network, dim, nonlinear, trainable):
        backbone, param = select_backbone(network)
        enc = nn.ModuleList([backbone, nn.AdaptiveAvgPool3d((1, 1, 1))])
        if nonlinear:
            enc.extend(_proj_modules(param['feature_size'], dim))
        return enc, param

    def _pairs(self):
        return list(zip(self.encoder_q.parameters(), self.encoder_k.parameters()))

    def _init_key_encoder(self):
        for q, k in self._pairs():
            k.data.copy_(q.data)
            k.requires_grad = False

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """k = m*k + (1-m)*q over the whole arena in one launch."""
        qs, ks = self.store, self.store_k
        dev = next(self.encoder_q.parameters()).device
        bb_q, bb_k = self.encoder_q[0], self.encoder_k[0]
        bb_q.prepare(dev)
        bb_k.prepare(dev)
        assert qs.total == ks.total, 'query / key arenas must have identical layouts'
        ops.call('dv_ema', ks.master, qs.master, ks.total, float(self.m), ks.dtype, ks.cc if ks.dtype != ops.DV_F32 else None)
        ks.mark_dirty(cast_done=True)

    @torch.no_grad()
    def _shuffle_rng_parity(self, n_local):
        """moco.py:142: every rank draws randperm(B*W) from the host RNG (then rank 0's is broadcast)."""
        import torch.distributed as dist
        torch.randperm(n_local * dist.get_world_size())

    @torch.no_grad()
    def _enqueue(self, queue, keys, ptr):
        n = keys.shape[0]
        assert self.K % n == 0
        queue[:, ptr:ptr + n] = keys.T


class MoCo_Naked(_MoCoBase):
    def __init__(self, network='s3d', dim=128, K=2048, m=0.999, T=0.07, distributed=True, nonlinear=True):
        super().__init__()
        self.dim, self.K, self.m, self.T, self.distributed, self.nonlinear = dim, K, m, T, distributed, nonlinear
        self.encoder_q, self.param = self._make_encoder(network, dim, nonlinear, True)
        self.encoder_k, _ = self._make_encoder(network, dim, nonlinear, False)
        self._init_key_encoder()
        self.register_buffer('queue', torch.randn(dim, K))
        self.queue = nn.functional.normalize(self.queue, dim=0)
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.criterion = nn.CrossEntropyLoss()
        hq = [(self.encoder_q[2], self.encoder_q[4])] if nonlinear else []
        hk = [(self.encoder_k[2], self.encoder_k[4])] if nonlinear else []
        self.store, hs = self._bind(self.encoder_q[0], hq)
        self.store_k = ParamStore()
        self.store_k.no_dgrad = True
        self.encoder_k[0].bind_store(self.store_k)
        objs = []
        for a, b in hk:
            DF.ProjectionHead.register(self.store_k, a, b)
            objs.append(DF.ProjectionHead(self.store_k, a, b))
        self._head_q = hs[0] if nonlinear else None
        self._head_k = objs[0] if nonlinear else None

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        """moco.py:109-126"""
        if self.distributed:
            keys = concat_all_gather(keys)
        ptr = int(self.queue_ptr)
        self._enqueue(self.queue, keys, ptr)
        self.queue_ptr[0] = (ptr + keys.shape[0]) % self.K

    def forward(self, block):
        B, N = block.shape[:2]
        assert N == 2
        self._sync_comm(self.encoder_q[0], self.encoder_k[0])
        pq = self.encoder_q[0].forward_pooled(block[:, 0])
        q = DF.l2_normalize(self._head_q(pq) if self._head_q is not None else pq).view(B, self.dim)
        train = q.requires_grad
        with torch.no_grad():
            if train:
                self._momentum_update_key_encoder()
            if self.distributed:
                self._shuffle_rng_parity(B)
            pk = self.encoder_k[0].forward_pooled(block[:, 1])
            k = DF.l2_normalize(self._head_k(pk) if self._head_k is not None else pk).view(B, self.dim)
        loss, logits, rank0 = DF.infonce(q, k, self.queue, self.T)
        labels = torch.zeros(B, dtype=torch.long, device=q.device)
        ret = {'clip_logits': logits, 'clip_labels': labels, 'clip_contrast_loss': loss, 'clip_rank0': rank0}
        if train:
            self._dequeue_and_enqueue(k)
        return ret


class MoCo_TimeSeriesV4(_MoCoBase):
    def __init__(self, network='s3d', dim=128, K=2048, m=0.999, T=0.07, distributed=True, nonlinear=True,
                 n_series=2, series_dim=64, series_T=0.07, aligned_T=0.07, mode='clip-sr-tc', args=None):
        super().__init__()
        self.args = args if args is not None else types.SimpleNamespace(shufflerank_theta=0.05)
        self.dim, self.K, self.m, self.T, self.distributed, self.nonlinear = dim, K, m, T, distributed, nonlinear
        self.n_series, self.series_dim, self.mode, self.series_T, self.aligned_T = n_series, series_dim, mode, series_T, aligned_T
        self.with_clip, self.with_sr, self.with_tc = 'clip' in mode, 'sr' in mode, 'tc' in mode
        self.encoder_q, self.param = self._make_encoder(network, dim, nonlinear, True)
        fs = self.param['feature_size']
        self.series_proj_head_q = nn.Sequential(*_proj_modules(fs, series_dim * n_series))
        self.encoder_k, _ = self._make_encoder(network, dim, nonlinear, False)
        self.series_proj_head_k = nn.Sequential(*_proj_modules(fs, series_dim * n_series))
        self._init_key_encoder()
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.register_buffer('queue', torch.randn(dim, K))
        self.queue = nn.functional.normalize(self.queue, dim=0)
        self.register_buffer('series_queue', torch.randn(series_dim * n_series, K))
        self.series_queue = nn.functional.normalize(self.series_queue.view(n_series, series_dim, K), dim=1) \
            .view(n_series * series_dim, K)
        self.criterion = nn.CrossEntropyLoss()
        assert nonlinear, 'the reference builds MoCo_TimeSeriesV4 with its projection heads'
        self.store, hs = self._bind(self.encoder_q[0], [(self.encoder_q[2], self.encoder_q[4]),
                                                        (self.series_proj_head_q[0], self.series_proj_head_q[2])])
        self._head_q, self._series_q = hs
        self.store_k = ParamStore()
        self.store_k.no_dgrad = True
        self.encoder_k[0].bind_store(self.store_k)
        self._head_k = self._series_k = None
        objs = []
        for a, b in ((self.encoder_k[2], self.encoder_k[4]), (self.series_proj_head_k[0], self.series_proj_head_k[2])):
            DF.ProjectionHead.register(self.store_k, a, b)
            objs.append(DF.ProjectionHead(self.store_k, a, b))
        self._head_k, self._series_k = objs

    def _pairs(self):
        return list(zip(self.encoder_q.parameters(), self.encoder_k.parameters())) + \
            list(zip(self.series_proj_head_q.parameters(), self.series_proj_head_k.parameters()))

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, series_keys):
        """moco.py:336-355"""
        if self.distributed:
            keys = concat_all_gather(keys)
            series_keys = concat_all_gather(series_keys)
        ptr = int(self.queue_ptr)
        self._enqueue(self.queue, keys, ptr)
        self._enqueue(self.series_queue, series_keys, ptr)
        self.queue_ptr[0] = (ptr + keys.shape[0]) % self.K

    def calc_clip_contrast_loss(self, q, k, queue, prefix='clip_'):
        """moco.py:426-438"""
        loss, logits, rank0 = DF.infonce(q, k, queue, self.T)
        labels = torch.zeros(q.size(0), dtype=torch.long, device=q.device)
        return {f'{prefix}logits': logits, f'{prefix}labels': labels, f'{prefix}contrast_loss': loss, f'{prefix}rank0': rank0}

    calc_contrast_loss = calc_clip_contrast_loss

    def calc_tc_contrast_loss(self, q, k, queue, prefix='tc_'):
        """moco.py:404-424.  mean_{i,j} q_i.n_j == <mean_i q_i, mean_j n_j>; with q' = tile(mean_i q_i)/s the
        [s*sd, K] series queue is contracted as stored: q'.queue[:, j] = <q_mean, n_mean_j>, q'.flat(k) = <q_mean, k_mean>."""
        B, s, sd = q.shape
        qm = DF.group_tile_div(DF.group_mean(q), s).view(B, s * sd)
        loss, logits, rank0 = DF.infonce(qm, k.reshape(B, s * sd), queue, self.aligned_T)
        labels = torch.zeros(B, dtype=torch.long, device=q.device)
        return {f'{prefix}logits': logits, f'{prefix}labels': labels, f'{prefix}contrast_loss': loss, f'{prefix}rank0': rank0}

    def calc_ranking_loss(self, features, n_views=2, prefix='ranking_', weight=1.):
        """moco.py:440-480: theta hard-coded 0.05, no clamp."""
        Bn, s, nv, dim = features.shape
        vm = features.permute(0, 2, 1, 3).reshape(Bn, nv * s, dim)
        loss, logits = DF.rank_margin(vm, s, 0.05, 0.0, weight)
        labels = torch.zeros(logits.size(0), dtype=torch.long, device=logits.device)
        return {f'{prefix}margin_logits': logits, f'{prefix}margin_labels': labels, f'{prefix}margin_contrast_loss': loss}

    def forward(self, block):
        ret = {}
        B, N, C, T, H, W = block.shape
        assert N == 3
        s, sd = self.n_series, self.series_dim
        bq, bk = self.encoder_q[0], self.encoder_k[0]
        self._sync_comm(bq, bk)
        pq = bq.forward_pooled(block[:, 0])
        q = DF.l2_normalize(self._head_q(pq)).view(B, self.dim)
        sf = DF.l2_normalize(self._series_q(pq).view(B, s, sd))
        train = q.requires_grad
        with torch.no_grad():
            if train:
                self._momentum_update_key_encoder()
            if self.distributed:
                self._shuffle_rng_parity(B)
            pk = bk.forward_pooled(block[:, 1])
            k = DF.l2_normalize(self._head_k(pk)).view(B, self.dim)
            sfk = DF.l2_normalize(self._series_k(pk).view(B, s, sd))
        ret.update(self.calc_clip_contrast_loss(q, k, self.queue, 'clip_'))
        if self.with_tc:
            ret.update(self.calc_tc_contrast_loss(sf, sfk, self.series_queue, 'tc_'))
        if train:
            self._dequeue_and_enqueue(k, sfk.reshape(B, s * sd))
        perm = np.array([np.random.permutation(s) for _ in range(B)])              # moco.py:544-546
        # one backbone pass over [aug_x1 ; shuffled aug_x1] (2B clips), exactly as the reference batches it
        aug = block[:, 2]
        dual = FrameBatch.cat([aug, aug]) if isinstance(aug, FrameBatch) else torch.cat([aug, aug], dim=0)
        ident = np.tile(np.arange(s), (B, 1))
        dp = bq.forward_pooled(dual, perm=np.concatenate([ident, perm], axis=0))
        dsf = DF.l2_normalize(self._series_q(dp).view(2 * B, s, sd))
        aug_sf, sh_sf = dsf[:B], dsf[B:]
        idx = torch.as_tensor(perm, dtype=torch.long, device=sh_sf.device).view(B, s, 1).expand_as(sh_sf)
        sh_sf = torch.scatter(sh_sf, 1, idx, sh_sf)
        ret.update(self.calc_ranking_loss(torch.stack([sf, sh_sf], dim=2), 2, 'unaug_ranking_', weight=0.5))
        ret.update(self.calc_ranking_loss(torch.stack([aug_sf, sh_sf], dim=2), 2, 'aug_ranking_', weight=0.5))
        return ret
