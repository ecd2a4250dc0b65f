"""S3D / S3D-G on the HIP engine (reference: backbone/s3dg.py:8-217).

Module and parameter names follow the reference (Conv_1a, Mixed_3b.branch1.1.conv2, gating_b0.fc, and the
blockN aliases of the stem) so that state_dict() keys are identical; the modules only hold parameters.
The network itself is `emit()`: per SepInception, the four branches write their outputs straight into
channel slices of one NDHWC buffer (no torch.cat), self-gating scales while writing the slice."""
import torch.nn as nn

from .base import HipBackbone, conv_geometry, emit_conv_bn, register_conv_bn


def _holder_init(conv, bn):
    conv.weight.data.normal_(mean=0, std=0.01)           # s3dg.py:20-22,51-56
    bn.weight.data.fill_(1)
    bn.bias.data.zero_()


class BasicConv3d(nn.Module):
    """holder for conv -> BN -> ReLU (s3dg.py:8-28)"""

    def __init__(self, cin, cout, kernel_size, stride, padding=0):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm3d(cout)
        _holder_init(self.conv, self.bn)

    def register(self, store, first=False):
        register_conv_bn(store, self.conv, self.bn, first)

    def emit(self, plan, x, out=None):
        return emit_conv_bn(plan, self.conv, self.bn, x, out=out)


class STConv3d(nn.Module):
    """holder for the separable pair 1xkxk -> BN -> ReLU -> kx1x1 -> BN -> ReLU (s3dg.py:30-65)"""

    def __init__(self, cin, cout, kernel_size, stride, padding=0):
        super().__init__()
        ts, ss = (stride[0], stride[-1]) if isinstance(stride, tuple) else (stride, stride)
        k, p = kernel_size, padding
        self.conv1 = nn.Conv3d(cin, cout, (1, k, k), (1, ss, ss), (0, p, p), bias=False)
        self.conv2 = nn.Conv3d(cout, cout, (k, 1, 1), (ts, 1, 1), (p, 0, 0), bias=False)
        self.bn1 = nn.BatchNorm3d(cout)
        self.bn2 = nn.BatchNorm3d(cout)
        _holder_init(self.conv1, self.bn1)
        _holder_init(self.conv2, self.bn2)

    def register(self, store, first=False):
        register_conv_bn(store, self.conv1, self.bn1, first)
        register_conv_bn(store, self.conv2, self.bn2)

    def emit(self, plan, x, out=None):
        y = emit_conv_bn(plan, self.conv1, self.bn1, x)
        return emit_conv_bn(plan, self.conv2, self.bn2, y, out=out)


class SelfGating(nn.Module):
    """holder for the gating Linear (s3dg.py:68-78)"""

    def __init__(self, c):
        super().__init__()
        self.fc = nn.Linear(c, c)

    def register(self, store):
        store.add_conv(self.fc.weight, need_dgrad=False)
        store.add_vec(self.fc.bias)


class SepInception(nn.Module):
    """s3dg.py:81-132"""

    def __init__(self, cin, out_planes, gating=False):
        super().__init__()
        o0, o1a, o1b, o2a, o2b, o3 = out_planes
        self.branch0 = nn.Sequential(BasicConv3d(cin, o0, 1, 1))
        self.branch1 = nn.Sequential(BasicConv3d(cin, o1a, 1, 1), STConv3d(o1a, o1b, 3, 1, 1))
        self.branch2 = nn.Sequential(BasicConv3d(cin, o2a, 1, 1), STConv3d(o2a, o2b, 3, 1, 1))
        self.branch3 = nn.Sequential(nn.MaxPool3d(3, 1, 1), BasicConv3d(cin, o3, 1, 1))
        self.widths = (o0, o1b, o2b, o3)
        self.out_channels = sum(self.widths)
        self.gating = gating
        if gating:
            self.gating_b0 = SelfGating(o0)
            self.gating_b1 = SelfGating(o1b)
            self.gating_b2 = SelfGating(o2b)
            self.gating_b3 = SelfGating(o3)

    def _gates(self):
        return (self.gating_b0, self.gating_b1, self.gating_b2, self.gating_b3) if self.gating else (None,) * 4

    def register(self, store):
        # the three 1x1x1 branch-entry convs read the same tensor: one [o0+o1a+o2a][Cin] GEMM
        self._entry = store.add_merged([self.branch0[0].conv.weight, self.branch1[0].conv.weight, self.branch2[0].conv.weight])
        self.branch0[0].register(store)
        self.branch1[0].register(store); self.branch1[1].register(store)
        self.branch2[0].register(store); self.branch2[1].register(store)
        self.branch3[1].register(store)
        if self.gating:
            # the four FC weights, then the four biases back to back: the bias gradient of the whole block is then one
            # column sum over [N, C_total] (the widths are multiples of 8, so slot offsets equal the concat offsets)
            for g in self._gates():
                store.add_conv(g.fc.weight, need_dgrad=False)
            for g in self._gates():
                store.add_vec(g.fc.bias)

    def emit(self, plan, x):
        """Level by level rather than branch by branch, so that the BatchNorms of one level share a single
        SyncBN statistics exchange: {b0, b1.0, b2.0, b3.1}, {b1.1.bn1, b2.1.bn1}, {b1.1.bn2, b2.1.bn2}."""
        cat = plan.act(x.N, x.T, x.H, x.W, self.out_channels)
        offs = [0]
        for w in self.widths:
            offs.append(offs[-1] + w)
        dst = [plan.slice(cat, offs[i], self.widths[i]) for i in range(4)]
        slot = plan.store.slot

        def conv(c, inp):
            return plan.conv(slot(c.weight), inp, *conv_geometry(c))

        b0, b1a, b2a, b3 = self.branch0[0], self.branch1[0], self.branch2[0], self.branch3[1]
        st1, st2 = self.branch1[1], self.branch2[1]
        mp = self.branch3[0]
        px = plan.maxpool(x, (mp.kernel_size,) * 3, (mp.stride,) * 3, (mp.padding,) * 3)
        ent = plan.store.add_merged([b0.conv.weight, b1a.conv.weight, b2a.conv.weight])
        raw = plan.conv(ent, x, (1, 1, 1), (1, 1, 1), (0, 0, 0))                    # [M, o0 | o1a | o2a]
        c0, c1, c2 = b0.conv.out_channels, b1a.conv.out_channels, b2a.conv.out_channels
        r0, r1, r2 = plan.slice(raw, 0, c0), plan.slice(raw, c0, c1), plan.slice(raw, c0 + c1, c2)
        for r in (r0, r1, r2):
            r.producer = raw.producer
        r3 = conv(b3.conv, px)
        _, y1, y2, _ = plan.bn_group([(b0.bn, r0, True, None, dst[0]), (b1a.bn, r1, True, None, None),
                                      (b2a.bn, r2, True, None, None), (b3.bn, r3, True, None, dst[3])])
        y1, y2 = plan.bn_group([(st1.bn1, conv(st1.conv1, y1), True, None, None),
                                (st2.bn1, conv(st2.conv1, y2), True, None, None)])
        plan.bn_group([(st1.bn2, conv(st1.conv2, y1), True, None, dst[1]),
                       (st2.bn2, conv(st2.conv2, y2), True, None, dst[2])])
        if self.gating:
            plan.gate_group([(g.fc, offs[i], self.widths[i]) for i, g in enumerate(self._gates())], cat)
        return cat


INCEPTION = (                                  # s3dg.py:163-192
    ('Mixed_3b', 192, [64, 96, 128, 16, 32, 32]), ('Mixed_3c', 256, [128, 128, 192, 32, 96, 64]),
    ('Mixed_4b', 480, [192, 96, 208, 16, 48, 64]), ('Mixed_4c', 512, [160, 112, 224, 24, 64, 64]),
    ('Mixed_4d', 512, [128, 128, 256, 24, 64, 64]), ('Mixed_4e', 512, [112, 144, 288, 32, 64, 64]),
    ('Mixed_4f', 528, [256, 160, 320, 32, 128, 128]), ('Mixed_5b', 832, [256, 160, 320, 32, 128, 128]),
    ('Mixed_5c', 832, [384, 192, 384, 48, 128, 128]))


class S3D(HipBackbone):
    feature_size = 1024

    def __init__(self, input_channel=3, gating=False, slow=False):
        super().__init__()
        assert input_channel == 3, 'the HIP ingest path is RGB-only'
        self.gating, self.slow = gating, slow
        mix = {n: (c, o) for n, c, o in INCEPTION}

        def inc(name):
            m = SepInception(*mix[name], gating=gating)
            setattr(self, name, m)
            return m

        self.Conv_1a = STConv3d(input_channel, 64, 7, (1, 2, 2) if slow else 2, 3)
        self.block1 = nn.Sequential(self.Conv_1a)
        self.MaxPool_2a = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        self.Conv_2b = BasicConv3d(64, 64, 1, 1)
        self.Conv_2c = STConv3d(64, 192, 3, 1, 1)
        self.block2 = nn.Sequential(self.MaxPool_2a, self.Conv_2b, self.Conv_2c)
        self.MaxPool_3a = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        self.block3 = nn.Sequential(self.MaxPool_3a, inc('Mixed_3b'), inc('Mixed_3c'))
        self.MaxPool_4a = nn.MaxPool3d(3, 2, 1)
        self.block4 = nn.Sequential(self.MaxPool_4a, *[inc(n) for n in ('Mixed_4b', 'Mixed_4c', 'Mixed_4d', 'Mixed_4e', 'Mixed_4f')])
        self.MaxPool_5a = nn.MaxPool3d(2, 2, 0)
        self.block5 = nn.Sequential(self.MaxPool_5a, inc('Mixed_5b'), inc('Mixed_5c'))

    def _sequence(self):
        for blk in (self.block1, self.block2, self.block3, self.block4, self.block5):
            for m in blk:
                yield m

    def register_params(self, store):
        first = True
        for m in self._sequence():
            if isinstance(m, (BasicConv3d, STConv3d)):
                m.register(store, first=first)
                first = False
            elif isinstance(m, SepInception):
                m.register(store)

    def emit(self, plan, x):
        def t3(v):
            return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)
        for m in self._sequence():
            if isinstance(m, nn.MaxPool3d):
                # the network is a chain here: the pool is the only reader of what precedes it (after a conv + BN + ReLU
                # -- MaxPool_2a, MaxPool_3a -- the engine then fuses the three)
                x = plan.maxpool(x, t3(m.kernel_size), t3(m.stride), t3(m.padding), sole_consumer=True)
            else:
                x = m.emit(plan, x)
        return x
