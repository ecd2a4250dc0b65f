"""Downstream `LinearClassifier` on the HIP engine, inference side (reference model/classifier.py:10-70; used by
classifier.py's test / retrieval passes `:657-738,787-995` and the 'last'-layer finetune): backbone in eval mode
(BatchNorm with running statistics) -> global average pool -> [L2 normalise] -> [BatchNorm1d] -> [Dropout = identity in
eval] -> Linear / MLP.  Same constructor, sub-module names (`backbone`, `final_bn`, `final_fc.N`), state_dict keys and
`(logit, feature)` return as the reference.  Training the head (dropout mask, cross-entropy) is the next SURVEY 8f row
and raises."""
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
    def _linear(self, x, lin, relu):
        st, lib = self.store, L.load()
        ws, bs = st.slot(lin.weight), st.slot(lin.bias)
        n, fin, fout = x.shape[0], ws.Cin, ws.Cout
        assert fin == ws.cin_pitch, 'head input widths must be multiples of 8'
        fp = cp8(fout)
        y = torch.empty(n, fp, dtype=torch.float32, device=x.device)
        ax = Act(x, n, 1, 1, 1, fin, fin, 0, DV_F32, fin)
        ay = Act(y, n, 1, 1, 1, fout, fp, 0, DV_F32, fp)
        d = ops.conv_desc(DV_F32, ax, ay, (1, 1, 1), (1, 1, 1), (0, 0, 0), flags=DV_BIAS | (DV_RELU if relu else 0))
        L.check(lib.dv_conv3d_fwd(C.byref(d), x.data_ptr(), st.w_master(ws), st.w_master(bs), y.data_ptr(), 0,
                                  ops.stream_ptr()), 'classifier linear')
        return y[:, :fout]

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
        if self.training:
            raise NotImplementedError('LinearClassifier training (dropout mask, cross-entropy, head gradients) is not built yet: '
                                      'call .eval() for the test / retrieval / feature-extraction passes')
        with torch.no_grad():
            feat3d = self.backbone.forward_pooled(block).contiguous()                   # [B, F] fp32
            if self.use_l2_norm:
                feat3d = DF.l2_normalize(feat3d)
            x = self._bn1d_eval(feat3d) if self.use_final_bn else feat3d
            for i, m in enumerate(self.final_fc):
                if isinstance(m, nn.Linear):
                    nxt = self.final_fc[i + 1] if i + 1 < len(self.final_fc) else None
                    x = self._linear(x.contiguous(), m, isinstance(nxt, nn.ReLU))
            return x, feat3d
