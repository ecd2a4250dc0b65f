"""Downstream `LinearClassifier` on the HIP engine (reference model/classifier.py:10-70; driven by classifier.py's
train / validate / test / retrieval passes `:422-498,501-543,657-738,787-995`): backbone -> global average pool ->
[L2 normalise] -> [BatchNorm1d] -> [Dropout] -> Linear / MLP.  Same constructor, sub-module names (`backbone`,
`final_bn`, `final_fc.N`), state_dict keys and `(logit, feature)` return as the reference.

Every sub-module follows ITS OWN `.training` flag, as in the reference's two modes (classifier.py:435-444):
  * `--train_what ft`   : `model.train()` -- backbone with batch statistics and gradients, dropout active;
  * `--train_what last` : `model.eval()` then `final_bn.train()` -- frozen eval-mode backbone (no autograd history),
                          dropout off, only the head trains.
The head's arithmetic (linear, BatchNorm1d, dropout multiply, and `DF.cross_entropy` for the criterion) runs in the HIP
library through torch.autograd.Function wrappers (dualvar_amd/functional.py); parameter gradients land in the arena."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib as L
from .. import functional as DF
from .. import ops
from ..backbone.select_backbone import select_backbone
from ..engine import ParamStore
from ..ops import DV_BIAS, DV_F32, DV_RELU, Act, cp8
from .simclr import _Objective


class LinearClassifier(_Objective):
    def __init__(self, num_class=101, network='resnet50', dropout=0.5, use_dropout=True, use_l2_norm=False,
                 use_final_bn=False, nonlinear=False, proj_dim=128):
        super().__init__()
        self.network, self.num_class, self.dropout = network, num_class, dropout
        self.use_dropout, self.use_l2_norm, self.use_final_bn = use_dropout, use_l2_norm, use_final_bn
        self.distributed = False
        self.backbone, self.param = select_backbone(network)
        F_ = self.param['feature_size']
        if use_final_bn:
            self.final_bn = nn.BatchNorm1d(F_)
            self.final_bn.weight.data.fill_(1)
            self.final_bn.bias.data.zero_()
        if use_dropout:
            self.final_fc = nn.Sequential(nn.Dropout(dropout), nn.Linear(F_, num_class))
        elif nonlinear:
            self.final_fc = nn.Sequential(nn.Linear(F_, proj_dim), nn.ReLU(), nn.Linear(proj_dim, num_class))
        else:
            self.final_fc = nn.Sequential(nn.Linear(F_, num_class))
        for name, prm in self.final_fc.named_parameters():          # classifier.py:64-70
            if 'bias' in name:
                nn.init.constant_(prm, 0.0)
            elif 'weight' in name:
                nn.init.normal_(prm, mean=0.0, std=0.01)
        self.store = ParamStore()
        self.backbone.bind_store(self.store)
        if use_final_bn:
            self.store.add_bn(self.final_bn)
        for m in self.final_fc:
            if isinstance(m, nn.Linear):
                self.store.add_conv(m.weight, need_dgrad=False)
                self.store.add_vec(m.bias)

    # ---- head pieces (fp32, [B, F] row-major)
    def _bn1d_eval(self, x):
        st, lib, bn = self.store, L.load(), self.final_bn
        n, Fdim = x.shape
        CP = cp8(Fdim)
        scale, shift = (torch.empty(CP, dtype=torch.float32, device=x.device) for _ in range(2))
        s = ops.stream_ptr()
        L.check(lib.dv_bn_eval_coeffs(st.w_master(st.slot(bn.weight)), st.w_master(st.slot(bn.bias)), bn.running_mean.data_ptr(),
                                      bn.running_var.data_ptr(), float(bn.eps), Fdim, scale.data_ptr(), shift.data_ptr(), s),
                'dv_bn_eval_coeffs')
        y = torch.empty_like(x)
        L.check(lib.dv_bn_apply(DV_F32, x.data_ptr(), Fdim, scale.data_ptr(), shift.data_ptr(), 0, 0, y.data_ptr(), Fdim, n, Fdim,
                                0, s), 'dv_bn_apply')
        return y

    def forward(self, block):
        if self.backbone.training:
            feat3d = self.backbone.forward_pooled(block)                                # [B, F] fp32, autograd-aware
        else:
            with torch.no_grad():
                feat3d = self.backbone.forward_pooled(block)
        feat3d = feat3d.contiguous()
        if self.use_l2_norm:
            feat3d = DF.l2_normalize(feat3d)
        x = feat3d
        if self.use_final_bn:
            if self.final_bn.training:
                x = DF.batchnorm1d_train(x, self.store, self.final_bn)
            else:
                if torch.is_grad_enabled() and x.requires_grad:
                    raise NotImplementedError('backward through an eval-mode final_bn is not built')
                x = self._bn1d_eval(x.detach())
        for i, m in enumerate(self.final_fc):
            if isinstance(m, nn.Dropout):
                if m.training:
                    x = DF.dropout(x, float(m.p))
            elif isinstance(m, nn.Linear):
                nxt = self.final_fc[i + 1] if i + 1 < len(self.final_fc) else None
                x = DF.linear(x, self.store, m, isinstance(nxt, nn.ReLU))
        return x, feat3d
