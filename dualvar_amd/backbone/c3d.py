"""C3D with BatchNorm (reference backbone/c3d.py:9-86): eight 3x3x3 convs WITH bias, each followed by BatchNorm3d +
ReLU, max-pools (1,2,2) then (2,2,2) x 3; [N,3,T,H,W] -> [N,512,T/8,H/16,W/16].  The conv kernels run bias-free: in
front of a BatchNorm the bias cancels in train mode (it only moves the running mean) and folds into the shift in eval
mode -- see Plan.bn(conv_bias=...); the bias parameters stay in the arena (weight decay acts on them as in the
reference; their gradient is exactly zero where the reference's is rounding noise)."""
import torch.nn as nn

from .base import HipBackbone, conv_geometry

CFG = (('1', 3, 64), ('2', 64, 128), ('3a', 128, 256), ('3b', 256, 256), ('4a', 256, 512), ('4b', 512, 512),
       ('5a', 512, 512), ('5b', 512, 512))


class C3D(HipBackbone):
    feature_size = 512
    stem_pad = 0            # 3x3x3 / stride 1 first conv: plain 4-channel (8-byte gather) ingest layout

    def __init__(self):
        super().__init__()
        for tag, cin, cout in CFG:
            setattr(self, 'conv' + tag, nn.Conv3d(cin, cout, 3, padding=1))
            setattr(self, 'bn' + tag, nn.BatchNorm3d(cout))
        self.pool1 = nn.MaxPool3d((1, 2, 2), (1, 2, 2))
        self.pool2 = nn.MaxPool3d(2, 2)
        self.pool3 = nn.MaxPool3d(2, 2)
        self.pool4 = nn.MaxPool3d(2, 2)

    def register_params(self, store):
        for i, (tag, cin, _) in enumerate(CFG):
            conv, bn = getattr(self, 'conv' + tag), getattr(self, 'bn' + tag)
            store.add_conv(conv.weight, cin_pitch=4 if cin == 3 else None, need_dgrad=i > 0)
            store.add_vec(conv.bias)
            store.add_bn(bn)

    def emit(self, plan, x):
        def cbr(tag, v):
            conv, bn = getattr(self, 'conv' + tag), getattr(self, 'bn' + tag)
            raw = plan.conv(plan.store.slot(conv.weight), v, *conv_geometry(conv))
            return plan.bn(bn, raw, relu=True, conv_bias=conv.bias)

        def pool(mp, v):
            t3 = lambda a: (a,) * 3 if isinstance(a, int) else tuple(a)     # noqa: E731
            return plan.maxpool(v, t3(mp.kernel_size), t3(mp.stride), t3(mp.padding))
        x = pool(self.pool1, cbr('1', x))
        x = pool(self.pool2, cbr('2', x))
        x = pool(self.pool3, cbr('3b', cbr('3a', x)))
        x = pool(self.pool4, cbr('4b', cbr('4a', x)))
        return cbr('5b', cbr('5a', x))
