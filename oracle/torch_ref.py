"""TEST INFRASTRUCTURE -- fp32 PyTorch-CPU restatement of DualVar's pretrain hot path.

This is the parity oracle (see oracle/__init__.py).  It is *not* the product and the
product never imports it.  It restates, device-agnostically and with the reference's
defects D1-D8 repaired, what these reference files compute:

    backbone/s3dg.py:8-217          -> S3D (+ unit(), st_unit(), Gate, Inception)
    backbone/r21d.py:11-266         -> R2Plus1D
    backbone/r3d.py:10-157          -> R3D
    backbone/resnet_2d3d.py:117-341 -> ResNet2d3d (Bottleneck2d/3d), the intended 'r50'
    backbone/select_backbone.py:7-31-> select_backbone
    utils/utils.py:75-92,321-338    -> calc_topk_accuracy, GatherLayer
    utils/transforms.py:57-63       -> normalize
    model/simclr.py:19-400          -> SimCLR_Naked, SimCLR_TimeSeriesV4
    model/moco.py:15-573            -> MoCo_Naked, MoCo_TimeSeriesV4

state_dict() key names and tensor shapes are identical to the reference's so that the
same procedural fill (oracle/procedural.py), keyed by name, initialises both.
Pinned by oracle/gen_golden.py (reference == oracle here) and tests/golden/*.npz.
"""
import math
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- utils
def normalize(vid, mean, std, channel=0):
    """utils/transforms.py:57-63 -- (x-mean)/std broadcast over `channel`."""
    shape = [1] * vid.dim()
    shape[channel] = -1
    mean = torch.as_tensor(mean, dtype=vid.dtype, device=vid.device).view(shape)
    std = torch.as_tensor(std, dtype=vid.dtype, device=vid.device).view(shape)
    return (vid - mean) / std


def calc_topk_accuracy(output, target, topk=(1,)):
    """utils/utils.py:75-92 -- fraction of rows whose target is within the top-k logits."""
    maxk = max(topk)
    n = target.size(0)
    pred = output.topk(maxk, 1, True, True)[1].t()
    hit = pred.eq(target.view(1, -1).expand_as(pred))
    return [hit[:k].reshape(-1).float().sum(0) * (1.0 / n) for k in topk]


class GatherLayer(torch.autograd.Function):
    """utils/utils.py:321-338 -- all_gather whose backward keeps only this rank's slice."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        out = [torch.zeros_like(x) for _ in range(dist.get_world_size())]
        dist.all_gather(out, x)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        (x,) = ctx.saved_tensors
        g = torch.zeros_like(x)
        g[:] = grads[dist.get_rank()]
        return g


# --------------------------------------------------------------------------- S3D / S3D-G
def _s3d_init(conv, *bns):
    conv.weight.data.normal_(mean=0, std=0.01)          # s3dg.py:20,51-52
    for bn in bns:
        bn.weight.data.fill_(1)
        bn.bias.data.zero_()


class BasicConv3d(nn.Module):
    """s3dg.py:8-28: conv(bias=False) -> BN -> ReLU."""

    def __init__(self, cin, cout, kernel_size, stride, padding=0):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm3d(cout)
        _s3d_init(self.conv, self.bn)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class STConv3d(nn.Module):
    """s3dg.py:30-65: 1xkxk conv -> BN -> ReLU -> kx1x1 conv -> BN -> ReLU."""

    def __init__(self, cin, cout, kernel_size, stride, padding=0):
        super().__init__()
        ts, ss = (stride[0], stride[-1]) if isinstance(stride, tuple) else (stride, stride)
        k, p = kernel_size, padding
        self.conv1 = nn.Conv3d(cin, cout, (1, k, k), (1, ss, ss), (0, p, p), bias=False)
        self.conv2 = nn.Conv3d(cout, cout, (k, 1, 1), (ts, 1, 1), (p, 0, 0), bias=False)
        self.bn1 = nn.BatchNorm3d(cout)
        self.bn2 = nn.BatchNorm3d(cout)
        _s3d_init(self.conv1, self.bn1)
        _s3d_init(self.conv2, self.bn2)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(x)))


class SelfGating(nn.Module):
    """s3dg.py:68-78: x * sigmoid(fc(mean_{T,H,W} x))."""

    def __init__(self, c):
        super().__init__()
        self.fc = nn.Linear(c, c)

    def forward(self, x):
        w = torch.sigmoid(self.fc(x.mean(dim=[2, 3, 4])))
        return w[:, :, None, None, None] * x


class SepInception(nn.Module):
    """s3dg.py:81-132."""

    def __init__(self, cin, out_planes, gating=False):
        super().__init__()
        o0, o1a, o1b, o2a, o2b, o3 = out_planes
        self.branch0 = nn.Sequential(BasicConv3d(cin, o0, 1, 1))
        self.branch1 = nn.Sequential(BasicConv3d(cin, o1a, 1, 1), STConv3d(o1a, o1b, 3, 1, 1))
        self.branch2 = nn.Sequential(BasicConv3d(cin, o2a, 1, 1), STConv3d(o2a, o2b, 3, 1, 1))
        self.branch3 = nn.Sequential(nn.MaxPool3d(3, 1, 1), BasicConv3d(cin, o3, 1, 1))
        self.out_channels = o0 + o1b + o2b + o3
        self.gating = gating
        if gating:
            self.gating_b0 = SelfGating(o0)
            self.gating_b1 = SelfGating(o1b)
            self.gating_b2 = SelfGating(o2b)
            self.gating_b3 = SelfGating(o3)

    def forward(self, x):
        ys = [self.branch0(x), self.branch1(x), self.branch2(x), self.branch3(x)]
        if self.gating:
            ys = [g(y) for g, y in zip((self.gating_b0, self.gating_b1, self.gating_b2, self.gating_b3), ys)]
        return torch.cat(ys, 1)


S3D_INCEPTION = {                              # s3dg.py:163-192
    'Mixed_3b': (192, [64, 96, 128, 16, 32, 32]),
    'Mixed_3c': (256, [128, 128, 192, 32, 96, 64]),
    'Mixed_4b': (480, [192, 96, 208, 16, 48, 64]),
    'Mixed_4c': (512, [160, 112, 224, 24, 64, 64]),
    'Mixed_4d': (512, [128, 128, 256, 24, 64, 64]),
    'Mixed_4e': (512, [112, 144, 288, 32, 64, 64]),
    'Mixed_4f': (528, [256, 160, 320, 32, 128, 128]),
    'Mixed_5b': (832, [256, 160, 320, 32, 128, 128]),
    'Mixed_5c': (832, [384, 192, 384, 48, 128, 128]),
}


class S3D(nn.Module):
    """s3dg.py:135-217.  Every stem module is registered under its own name *and* inside
    blockN (state_dict holds both aliases)."""

    def __init__(self, input_channel=3, gating=False, slow=False):
        super().__init__()
        self.gating = gating
        self.Conv_1a = STConv3d(input_channel, 64, 7, (1, 2, 2) if slow else 2, 3)
        self.block1 = nn.Sequential(self.Conv_1a)
        self.MaxPool_2a = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        self.Conv_2b = BasicConv3d(64, 64, 1, 1)
        self.Conv_2c = STConv3d(64, 192, 3, 1, 1)
        self.block2 = nn.Sequential(self.MaxPool_2a, self.Conv_2b, self.Conv_2c)
        self.MaxPool_3a = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        for n in ('Mixed_3b', 'Mixed_3c'):
            setattr(self, n, SepInception(*S3D_INCEPTION[n], gating=gating))
        self.block3 = nn.Sequential(self.MaxPool_3a, self.Mixed_3b, self.Mixed_3c)
        self.MaxPool_4a = nn.MaxPool3d(3, 2, 1)
        for n in ('Mixed_4b', 'Mixed_4c', 'Mixed_4d', 'Mixed_4e', 'Mixed_4f'):
            setattr(self, n, SepInception(*S3D_INCEPTION[n], gating=gating))
        self.block4 = nn.Sequential(self.MaxPool_4a, self.Mixed_4b, self.Mixed_4c, self.Mixed_4d,
                                    self.Mixed_4e, self.Mixed_4f)
        self.MaxPool_5a = nn.MaxPool3d(2, 2, 0)
        for n in ('Mixed_5b', 'Mixed_5c'):
            setattr(self, n, SepInception(*S3D_INCEPTION[n], gating=gating))
        self.block5 = nn.Sequential(self.MaxPool_5a, self.Mixed_5b, self.Mixed_5c)

    def forward(self, x):
        for blk in (self.block1, self.block2, self.block3, self.block4, self.block5):
            x = blk(x)
        return x


# --------------------------------------------------------------------------- R(2+1)D / R3D
def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


class R21DConv(nn.Module):
    """r21d.py:11-70 (SpatioTemporalConv): 1xkxk -> BN -> ReLU -> kx1x1, mid-channels :47-49."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        k, s, p = _triple(kernel_size), _triple(stride), _triple(padding)
        mid = int(math.floor((k[0] * k[1] * k[2] * cin * cout) / (k[1] * k[2] * cin + k[0] * cout)))
        self.spatial_conv = nn.Conv3d(cin, mid, (1, k[1], k[2]), (1, s[1], s[2]), (0, p[1], p[2]), bias=False)
        self.bn = nn.BatchNorm3d(mid)
        self.temporal_conv = nn.Conv3d(mid, cout, (k[0], 1, 1), (s[0], 1, 1), (p[0], 0, 0), bias=False)

    def forward(self, x):
        return self.temporal_conv(F.relu(self.bn(self.spatial_conv(x))))


class R3DConv(nn.Module):
    """r3d.py:10-38 (SpatioTemporalConv): one full conv."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        self.temporal_spatial_conv = nn.Conv3d(cin, cout, _triple(kernel_size), _triple(stride),
                                               _triple(padding), bias=False)

    def forward(self, x):
        return self.temporal_spatial_conv(x)


class ResBlock(nn.Module):
    """r21d.py:73-122 / r3d.py:41-89 (SpatioTemporalResBlock)."""

    def __init__(self, conv_t, cin, cout, kernel_size, downsample=False):
        super().__init__()
        self.downsample = downsample
        pad = kernel_size // 2
        if downsample:
            self.downsampleconv = conv_t(cin, cout, 1, stride=2)
            self.downsamplebn = nn.BatchNorm3d(cout)
            self.conv1 = conv_t(cin, cout, kernel_size, padding=pad, stride=2)
        else:
            self.conv1 = conv_t(cin, cout, kernel_size, padding=pad)
        self.bn1 = nn.BatchNorm3d(cout)
        self.conv2 = conv_t(cout, cout, kernel_size, padding=pad)
        self.bn2 = nn.BatchNorm3d(cout)

    def forward(self, x):
        res = F.relu(self.bn1(self.conv1(x)))
        res = self.bn2(self.conv2(res))
        if self.downsample:
            x = self.downsamplebn(self.downsampleconv(x))
        return F.relu(x + res)


class ResLayer(nn.Module):
    """r21d.py:176-211 / r3d.py:92-123 (SpatioTemporalResLayer)."""

    def __init__(self, conv_t, cin, cout, kernel_size, layer_size, downsample=False):
        super().__init__()
        self.block1 = ResBlock(conv_t, cin, cout, kernel_size, downsample)
        self.blocks = nn.ModuleList([ResBlock(conv_t, cout, cout, kernel_size) for _ in range(layer_size - 1)])

    def forward(self, x):
        x = self.block1(x)
        for b in self.blocks:
            x = b(x)
        return x


class _ResNet18ish(nn.Module):
    def __init__(self, conv_t, layer_sizes):
        super().__init__()
        self.conv1 = conv_t(3, 64, (3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3))
        self.bn1 = nn.BatchNorm3d(64)
        self.conv2 = ResLayer(conv_t, 64, 64, 3, layer_sizes[0])
        self.conv3 = ResLayer(conv_t, 64, 128, 3, layer_sizes[1], downsample=True)
        self.conv4 = ResLayer(conv_t, 128, 256, 3, layer_sizes[2], downsample=True)
        self.conv5 = ResLayer(conv_t, 256, 512, 3, layer_sizes[3], downsample=True)

    def forward(self, x, ret_frame_feature=False, multi_level=False):
        x = F.relu(self.bn1(self.conv1(x)))
        feats = []
        for layer in (self.conv2, self.conv3, self.conv4, self.conv5):
            x = layer(x)
            feats.append(x)
        if not ret_frame_feature:
            return x
        return (x, feats) if multi_level else (x, feats[0])


class R2Plus1DNet(_ResNet18ish):
    """r21d.py:214-266, default layer_sizes (1,1,1,1) = the paper's 14.4 M-param net (D10)."""

    def __init__(self, layer_sizes=(1, 1, 1, 1)):
        super().__init__(R21DConv, layer_sizes)


class R3DNet(_ResNet18ish):
    """r3d.py:126-157."""

    def __init__(self, layer_sizes=(1, 1, 1, 1)):
        super().__init__(R3DConv, layer_sizes)


# --------------------------------------------------------------------------- ResNet2d3d
class _Bottleneck(nn.Module):
    """resnet_2d3d.py:117-200: conv1 is 3x1x1 in the 3d block, 1x1x1 in the 2d block."""
    expansion = 4
    temporal = False

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_final_relu=True):
        super().__init__()
        self.use_final_relu = use_final_relu
        if self.temporal:
            self.conv1 = nn.Conv3d(inplanes, planes, (3, 1, 1), padding=(1, 0, 0), bias=False)
        else:
            self.conv1 = nn.Conv3d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, (1, 3, 3), (1, stride, stride), (0, 1, 1), bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        out = out + (x if self.downsample is None else self.downsample(x))
        return F.relu(out) if self.use_final_relu else out


class Bottleneck2d(_Bottleneck):
    temporal = False


class Bottleneck3d(_Bottleneck):
    temporal = True


class ResNet2d3d(nn.Module):
    """resnet_2d3d.py:272-341 (the constructor that works; D7)."""

    def __init__(self, block, layers, input_channel=3):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv3d(input_channel, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        self.maxpool = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        if not isinstance(block, list):
            block = [block] * 4
        self.layer1 = self._make_layer(block[0], 64, layers[0])
        self.layer2 = self._make_layer(block[1], 128, layers[1], stride=(1, 2, 2))
        self.layer3 = self._make_layer(block[2], 256, layers[2], stride=(1, 2, 2))
        self.layer4 = self._make_layer(block[3], 512, layers[3], stride=(1, 2, 2), is_final=True)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, is_final=False):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            if isinstance(stride, int):
                cstride = (1, stride, stride) if block is Bottleneck2d else stride
            else:
                cstride, stride = stride, stride[-1]
            downsample = nn.Sequential(
                nn.Conv3d(self.inplanes, planes * block.expansion, 1, cstride, bias=False),
                nn.BatchNorm3d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        if is_final:
            layers += [block(self.inplanes, planes) for _ in range(1, blocks - 1)]
            layers.append(block(self.inplanes, planes, use_final_relu=False))
        else:
            layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
        return F.relu(x)


class BasicBlock2d(nn.Module):
    """resnet_2d3d.py:45-78: two 1x3x3 convs (+BN), identity / 1x1x1-conv shortcut, optional final ReLU."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_final_relu=True):
        super().__init__()
        self.use_final_relu = use_final_relu
        self.conv1 = nn.Conv3d(inplanes, planes, (1, 3, 3), (1, stride, stride), (0, 1, 1), bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, (1, 3, 3), 1, (0, 1, 1), bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.downsample = downsample

    def forward(self, x):
        out = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        out = out + (x if self.downsample is None else self.downsample(x))
        return F.relu(out) if self.use_final_relu else out


class ResNet2d3dFull(nn.Module):
    """resnet_2d3d.py:203-270 `ResNet2d3d_full` as `r2d3d18()` instantiates it (:352-356): BasicBlock2d x [2,2,2,2], stem
    1x7x7 / (1,2,2), layer4 at 256 planes whose last block has no final ReLU -- and, unlike ResNet2d3d, no ReLU after it
    either (`forward` :259-270)."""

    def __init__(self, block=BasicBlock2d, layers=(2, 2, 2, 2)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv3d(3, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        self.maxpool = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 256, layers[3], stride=2, is_final=True)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, is_final=False):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv3d(self.inplanes, planes * block.expansion, 1, (1, stride, stride), bias=False),
                                       nn.BatchNorm3d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        if is_final:
            layers += [block(self.inplanes, planes) for _ in range(1, blocks - 1)]
            layers.append(block(self.inplanes, planes, use_final_relu=False))
        else:
            layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
        return x


class C3D(nn.Module):
    """c3d.py:9-86: eight 3x3x3 convs WITH bias, each followed by BatchNorm3d + ReLU; pools (1,2,2) then (2,2,2) x 3."""

    def __init__(self):
        super().__init__()
        cfg = (('1', 3, 64), ('2', 64, 128), ('3a', 128, 256), ('3b', 256, 256), ('4a', 256, 512), ('4b', 512, 512),
               ('5a', 512, 512), ('5b', 512, 512))
        for tag, cin, cout in cfg:
            setattr(self, 'conv' + tag, nn.Conv3d(cin, cout, 3, padding=1))
            setattr(self, 'bn' + tag, nn.BatchNorm3d(cout))
        self.pool1 = nn.MaxPool3d((1, 2, 2), (1, 2, 2))
        self.pool2 = nn.MaxPool3d(2, 2)
        self.pool3 = nn.MaxPool3d(2, 2)
        self.pool4 = nn.MaxPool3d(2, 2)

    def forward(self, x):
        def cbr(tag, v):
            return F.relu(getattr(self, 'bn' + tag)(getattr(self, 'conv' + tag)(v)))
        x = self.pool1(cbr('1', x))
        x = self.pool2(cbr('2', x))
        x = self.pool3(cbr('3b', cbr('3a', x)))
        x = self.pool4(cbr('4b', cbr('4a', x)))
        return cbr('5b', cbr('5a', x))


def select_backbone(network, first_channel=3):
    """select_backbone.py:7-31; 'r50' built as the survey's D7 repair."""
    param = {'feature_size': 1024}
    if network == 's3d':
        model = S3D(input_channel=first_channel)
    elif network == 's3dg':
        model = S3D(input_channel=first_channel, gating=True)
    elif network == 'r50':
        param['feature_size'] = 2048
        model = ResNet2d3d([Bottleneck2d, Bottleneck2d, Bottleneck3d, Bottleneck3d], [3, 4, 6, 3], first_channel)
    elif network == 'r21d':
        param['feature_size'] = 512
        model = R2Plus1DNet()
    elif network == 'r3d':
        param['feature_size'] = 512
        model = R3DNet()
    elif network == 'r2d3d18':
        param['feature_size'] = 256
        model = ResNet2d3dFull()
    elif network == 'c3d':
        param['feature_size'] = 512
        model = C3D()
    else:
        raise NotImplementedError(network)
    return model, param


# --------------------------------------------------------------------------- objectives
def _proj_head(cin, cout):
    return [nn.Conv3d(cin, cin, 1, bias=True), nn.ReLU(), nn.Conv3d(cin, cout, 1, bias=True)]


def ntxent_from_similarity(sim, row_index, n_per_view, temperature):
    """The masked-select / cat sequence of simclr.py:66-91 (also :196-221, :299-329).

    sim: [R, 2N] similarities of R rows against the 2N view-major columns; row r is global
    row row_index[r].  Returns logits [R, 2N-1] = [positive, negatives in column order] / T.
    """
    R, C = sim.shape
    cols = torch.arange(C, device=sim.device)
    rows = row_index.to(sim.device)
    self_mask = cols[None, :] == rows[:, None]
    pos_mask = (cols[None, :] % n_per_view) == (rows[:, None] % n_per_view)
    pos_mask = pos_mask & ~self_mask
    pos = sim[pos_mask].view(R, -1)
    neg = sim[~(pos_mask | self_mask)].view(R, -1)
    return torch.cat([pos, neg], dim=1) / temperature


class _SimCLRBase(nn.Module):
    def _encode(self, x, upto=None):
        feats = x
        pooled = None
        for i, mod in enumerate(self.encoder_q):
            if upto is not None and i > upto:
                break
            feats = mod(feats)
            if i == 1:
                pooled = feats
        return feats, pooled

    def _gather(self, t):
        return torch.cat(GatherLayer.apply(t), dim=0) if self.distributed else t

    def calc_clip_contrast_loss(self, features, n_views=2, prefix='clip_'):
        """simclr.py:56-99 / :183-229.  features [B, 2, dim], already L2-normalised."""
        B, nv, dim = features.shape
        features = self._gather(features)
        N = features.size(0)
        f = features.permute(1, 0, 2).reshape(nv * N, dim)             # view-major (D2 repaired)
        logits = ntxent_from_similarity(f @ f.t(), torch.arange(nv * N), N, self.T)
        labels = torch.zeros(logits.size(0), dtype=torch.long, device=logits.device)
        return {f'{prefix}logits': logits, f'{prefix}labels': labels,
                f'{prefix}contrast_loss': F.cross_entropy(logits, labels)}

    calc_contrast_loss = calc_clip_contrast_loss                       # D1 alias


class SimCLR_Naked(_SimCLRBase):
    """simclr.py:19-127."""

    def __init__(self, network='s3d', dim=128, T=0.07, distributed=True, nonlinear=True):
        super().__init__()
        self.dim, self.T, self.distributed, self.nonlinear = dim, T, distributed, nonlinear
        backbone, self.param = select_backbone(network)
        fs = self.param['feature_size']
        self.encoder_q = nn.ModuleList([backbone, nn.AdaptiveAvgPool3d((1, 1, 1))])
        if nonlinear:
            self.encoder_q.extend(_proj_head(fs, dim))
        self.criterion = nn.CrossEntropyLoss()

    def forward(self, block):
        B, nv = block.shape[:2]
        assert nv == 2
        feats, _ = self._encode(block.reshape(-1, *block.shape[2:]))
        feats = F.normalize(feats, dim=1).reshape(B, nv, self.dim)
        return self.calc_clip_contrast_loss(feats, nv, 'clip_')


class SimCLR_TimeSeriesV4(_SimCLRBase):
    """simclr.py:130-400."""

    def __init__(self, network='s3d', dim=128, T=0.07, distributed=True, nonlinear=True, n_series=2,
                 series_dim=64, series_T=0.07, aligned_T=0.07, mode='clip-sr-tc', args=None):
        super().__init__()
        self.args = args if args is not None else types.SimpleNamespace(shufflerank_theta=0.05)
        self.dim, self.T, self.distributed, self.nonlinear = dim, T, distributed, nonlinear
        self.n_series, self.series_dim, self.series_T, self.aligned_T, self.mode = \
            n_series, series_dim, series_T, aligned_T, mode
        self.with_clip, self.with_sr, self.with_tc = 'clip' in mode, 'sr' in mode, 'tc' in mode
        backbone, self.param = select_backbone(network)
        fs = self.param['feature_size']
        self.encoder_q = nn.ModuleList([backbone, nn.AdaptiveAvgPool3d((1, 1, 1))])
        if nonlinear and self.with_clip:
            self.encoder_q.extend(_proj_head(fs, dim))
        self.criterion = nn.CrossEntropyLoss()
        self.series_proj_head = nn.Sequential(*_proj_head(fs, series_dim * n_series))

    def calc_ranking_loss(self, features, n_views=2, prefix='ranking_', weight=1.):
        """simclr.py:231-278.  features [Bn, s, 2, sd]."""
        return ranking_loss(features, self.n_series, n_views, prefix, weight,
                            theta=self.args.shufflerank_theta, clip=5.0)

    def calc_tc_contrast_loss(self, features, prefix='tc_'):
        """simclr.py:280-337.  features [B, 2, s, sd]; rows = this rank's 2B entries."""
        B, nv, s, sd = features.shape
        rank, world = 0, 1
        if self.distributed:
            features = self._gather(features)
            rank, world = dist.get_rank(), dist.get_world_size()
        N = features.size(0)
        n = N // world
        base = n * rank
        rows = features[base:base + n].permute(1, 0, 2, 3).reshape(nv * n, s, sd)
        cols = features.permute(1, 0, 2, 3).reshape(nv * N, s, sd)
        sim = torch.matmul(rows.unsqueeze(1), cols.unsqueeze(0).transpose(3, 2)).mean(dim=(2, 3))
        row_index = (torch.arange(base, base + n)[None, :] + torch.arange(nv)[:, None] * N).reshape(-1)
        logits = ntxent_from_similarity(sim, row_index, N, self.aligned_T)
        labels = torch.zeros(logits.size(0), dtype=torch.long, device=logits.device)
        return {f'{prefix}logits': logits, f'{prefix}labels': labels,
                f'{prefix}contrast_loss': F.cross_entropy(logits, labels)}

    def forward(self, block):
        block = block.contiguous()
        B, NV, C, T, H, W = block.shape
        assert NV == 3
        s, sd = self.n_series, self.series_dim
        feats, pooled = self._encode(block.reshape(-1, C, T, H, W))
        ret = {}
        if self.with_clip:
            feats = F.normalize(feats, dim=1).reshape(B, NV, self.dim)[:, :2].contiguous()
            ret.update(self.calc_clip_contrast_loss(feats, 2))
        series = F.normalize(self.series_proj_head(pooled).reshape(B, NV, s, sd), dim=3)
        if self.with_tc:
            ret.update(self.calc_tc_contrast_loss(series[:, :2].contiguous()))
        if self.with_sr:
            orig = series[:, [0, 2]].contiguous()
            perm = torch.as_tensor(np.array([np.random.permutation(s) for _ in range(B)]),
                                   dtype=torch.long, device=block.device)          # simclr.py:378-381
            x = block[:, 2].reshape(B, C, s, T // s, H, W)
            shuffled = torch.gather(x, 2, perm.view(B, 1, s, 1, 1, 1).expand_as(x)).reshape(B, C, T, H, W)
            _, sp = self._encode(shuffled, upto=1)
            sf = self.series_proj_head(sp).reshape(B, s, sd)
            sf = torch.scatter(sf, 1, perm.view(B, s, 1).expand_as(sf), sf)          # un-permute :389-392
            sf = F.normalize(sf, dim=2)
            ret.update(self.calc_ranking_loss(torch.stack([orig[:, 0], sf], dim=2), 2, 'aug_ranking_', 0.5))
            ret.update(self.calc_ranking_loss(torch.stack([orig[:, 1], sf], dim=2), 2, 'unaug_ranking_', 0.5))
        return ret


def ranking_loss(features, n_series, n_views, prefix, weight, theta, clip):
    """simclr.py:231-278 (theta from args, clip 5) and moco.py:440-480 (theta .05, no clip)."""
    Bn, s, nv, dim = features.shape
    assert s == n_series and nv == n_views
    f = features.permute(0, 2, 1, 3).reshape(Bn, nv * s, dim)
    sim = torch.bmm(f, f.transpose(2, 1))
    idx = torch.arange(nv * s, device=f.device)
    eye = idx[:, None] == idx[None, :]
    corr = ((idx[:, None] % s) == (idx[None, :] % s)) & ~eye
    left = ~(eye | corr)
    hi = sim[:, corr].view(Bn, nv * s, 1)
    lo = sim[:, left].view(Bn, nv * s, nv * s - 2)
    z = (lo - hi) / theta
    if clip is not None:
        z = z.clip(max=clip)
    loss = weight * torch.log(1 + torch.exp(z)).mean()
    logits = torch.cat([hi, lo], dim=2).view(-1, nv * s - 1)
    labels = torch.zeros(logits.size(0), dtype=torch.long, device=f.device)
    return {f'{prefix}margin_logits': logits, f'{prefix}margin_labels': labels,
            f'{prefix}margin_contrast_loss': loss}


@torch.no_grad()
def concat_all_gather(t):
    """moco.py:14-25."""
    out = [torch.ones_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t, async_op=False)
    return torch.cat(out, dim=0)


class _MoCoBase(nn.Module):
    def _pairs(self):
        return list(zip(self.encoder_q.parameters(), self.encoder_k.parameters()))

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """moco.py:104-107 / :329-334."""
        for q, k in self._pairs():
            k.data = k.data * self.m + q.data * (1. - self.m)

    @torch.no_grad()
    def _batch_shuffle_ddp(self, x):
        """moco.py:129-155."""
        n_this = x.shape[0]
        xg = concat_all_gather(x)
        idx = torch.randperm(xg.shape[0]).to(x.device)
        dist.broadcast(idx, src=0)
        unshuffle = torch.argsort(idx)
        return xg[idx.view(xg.shape[0] // n_this, -1)[dist.get_rank()]], unshuffle

    @torch.no_grad()
    def _batch_unshuffle_ddp(self, x, unshuffle):
        """moco.py:157-173."""
        n_this = x.shape[0]
        xg = concat_all_gather(x)
        return xg[unshuffle.view(xg.shape[0] // n_this, -1)[dist.get_rank()]]

    def _infonce(self, pos, neg, T, prefix):
        logits = torch.cat([pos, neg], dim=1) / T
        labels = torch.zeros(logits.size(0), dtype=torch.long, device=logits.device)
        return {f'{prefix}logits': logits, f'{prefix}labels': labels,
                f'{prefix}contrast_loss': F.cross_entropy(logits, labels)}


def _run_modlist(mods, x):
    pooled = None
    for i, mod in enumerate(mods):
        x = mod(x)
        if i == 1:
            pooled = x
    return x, pooled


class MoCo_Naked(_MoCoBase):
    """moco.py:28-239."""

    def __init__(self, network='s3d', dim=128, K=2048, m=0.999, T=0.07, distributed=True, nonlinear=True):
        super().__init__()
        self.dim, self.K, self.m, self.T, self.distributed, self.nonlinear = dim, K, m, T, distributed, nonlinear
        for name in ('encoder_q', 'encoder_k'):
            backbone, self.param = select_backbone(network)
            enc = nn.ModuleList([backbone, nn.AdaptiveAvgPool3d((1, 1, 1))])
            if nonlinear:
                enc.extend(_proj_head(self.param['feature_size'], dim))
            setattr(self, name, enc)
        for q, k in self._pairs():
            k.data.copy_(q.data)
            k.requires_grad = False
        self.register_buffer('queue', torch.randn(dim, K))
        self.queue = F.normalize(self.queue, dim=0)
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.criterion = nn.CrossEntropyLoss()

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        """moco.py:109-126."""
        if self.distributed:
            keys = concat_all_gather(keys)
        n = keys.shape[0]
        ptr = int(self.queue_ptr)
        assert self.K % n == 0
        self.queue[:, ptr:ptr + n] = keys.T
        self.queue_ptr[0] = (ptr + n) % self.K

    def forward(self, block):
        B, N = block.shape[:2]
        assert N == 2
        x1, x2 = block[:, 0].contiguous(), block[:, 1].contiguous()
        q, _ = _run_modlist(self.encoder_q, x1)
        q = F.normalize(q, dim=1).view(B, self.dim)
        train = q.requires_grad
        with torch.no_grad():
            if train:
                self._momentum_update_key_encoder()
            if self.distributed:
                x2, unshuffle = self._batch_shuffle_ddp(x2)
            k, _ = _run_modlist(self.encoder_k, x2)
            k = F.normalize(k, dim=1)
            if self.distributed:
                k = self._batch_unshuffle_ddp(k, unshuffle)
        k = k.view(B, self.dim)
        pos = torch.einsum('nc,nc->n', [q, k]).unsqueeze(-1)
        neg = torch.einsum('nc,ck->nk', [q, self.queue.clone().detach()])
        ret = self._infonce(pos, neg, self.T, 'clip_')
        if train:
            self._dequeue_and_enqueue(k)
        return ret


class MoCo_TimeSeriesV4(_MoCoBase):
    """moco.py:242-573."""

    def __init__(self, network='s3d', dim=128, K=2048, m=0.999, T=0.07, distributed=True, nonlinear=True,
                 n_series=2, series_dim=64, series_T=0.07, aligned_T=0.07, mode='clip-sr-tc', args=None):
        super().__init__()
        self.dim, self.K, self.m, self.T, self.distributed, self.nonlinear = dim, K, m, T, distributed, nonlinear
        self.n_series, self.series_dim, self.mode, self.series_T, self.aligned_T = \
            n_series, series_dim, mode, series_T, aligned_T
        self.with_clip, self.with_sr, self.with_tc = 'clip' in mode, 'sr' in mode, 'tc' in mode
        for tag in ('q', 'k'):
            backbone, self.param = select_backbone(network)
            fs = self.param['feature_size']
            enc = nn.ModuleList([backbone, nn.AdaptiveAvgPool3d((1, 1, 1))])
            if nonlinear:
                enc.extend(_proj_head(fs, dim))
            setattr(self, f'encoder_{tag}', enc)
            setattr(self, f'series_proj_head_{tag}', nn.Sequential(*_proj_head(fs, series_dim * n_series)))
            if tag == 'q':
                # registration order in the reference: encoder_q, series_proj_head_q, encoder_k, ..._k
                pass
        for q, k in self._pairs():
            k.data.copy_(q.data)
            k.requires_grad = False
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.register_buffer('queue', torch.randn(dim, K))
        self.queue = F.normalize(self.queue, dim=0)
        self.register_buffer('series_queue', torch.randn(series_dim * n_series, K))
        self.series_queue = F.normalize(self.series_queue.view(n_series, series_dim, K), dim=1) \
            .view(n_series * series_dim, K)
        self.criterion = nn.CrossEntropyLoss()

    def _pairs(self):
        return list(zip(self.encoder_q.parameters(), self.encoder_k.parameters())) + \
            list(zip(self.series_proj_head_q.parameters(), self.series_proj_head_k.parameters()))

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, series_keys):
        """moco.py:336-355."""
        if self.distributed:
            keys = concat_all_gather(keys)
            series_keys = concat_all_gather(series_keys)
        n = keys.shape[0]
        ptr = int(self.queue_ptr)
        assert self.K % n == 0
        self.queue[:, ptr:ptr + n] = keys.T
        self.series_queue[:, ptr:ptr + n] = series_keys.T
        self.queue_ptr[0] = (ptr + n) % self.K

    def calc_tc_contrast_loss(self, q, k, queue, prefix='tc_'):
        """moco.py:404-424."""
        B, s, sd = q.shape
        neg = queue.clone().detach().T.contiguous().view(self.K, s, sd)
        pos_l = torch.matmul(q, k.transpose(2, 1).contiguous()).mean(dim=(1, 2)).unsqueeze(1)
        neg_l = torch.matmul(q.unsqueeze(1), neg.transpose(2, 1).contiguous()).mean(dim=(2, 3))
        return self._infonce(pos_l, neg_l, self.aligned_T, prefix)

    def calc_clip_contrast_loss(self, q, k, queue, prefix='clip_'):
        """moco.py:426-438."""
        pos = torch.einsum('nc,nc->n', [q, k]).unsqueeze(-1)
        neg = torch.einsum('nc,ck->nk', [q, queue.clone().detach()])
        return self._infonce(pos, neg, self.T, prefix)

    calc_contrast_loss = calc_clip_contrast_loss                       # D1 alias

    def calc_ranking_loss(self, features, n_views=2, prefix='ranking_', weight=1.):
        """moco.py:440-480 (theta hard-coded .05, no clip: D12)."""
        return ranking_loss(features, self.n_series, n_views, prefix, weight, theta=0.05, clip=None)

    def forward(self, block):
        ret = {}
        B, N, C, T, H, W = block.shape
        assert N == 3
        s, sd = self.n_series, self.series_dim
        x1, x2, aug = (block[:, i].contiguous() for i in range(3))
        q, pooled_q = _run_modlist(self.encoder_q, x1)
        q = F.normalize(q, dim=1).view(B, self.dim)
        sf = F.normalize(self.series_proj_head_q(pooled_q).view(B, s, sd), dim=2)
        train = q.requires_grad
        with torch.no_grad():
            if train:
                self._momentum_update_key_encoder()
            if self.distributed:
                x2, unshuffle = self._batch_shuffle_ddp(x2)
            k, pooled_k = _run_modlist(self.encoder_k, x2)
            k = F.normalize(k, dim=1)
            sfk = F.normalize(self.series_proj_head_k(pooled_k).view(B, s, sd), dim=2).view(B, s * sd)
            if self.distributed:
                k = self._batch_unshuffle_ddp(k, unshuffle)
                sfk = self._batch_unshuffle_ddp(sfk, unshuffle)
        k = k.view(B, self.dim)
        ret.update(self.calc_clip_contrast_loss(q, k, self.queue, 'clip_'))
        sfk = sfk.view(B, s, sd)
        if self.with_tc:
            ret.update(self.calc_tc_contrast_loss(sf, sfk, self.series_queue, 'tc_'))
        if train:
            self._dequeue_and_enqueue(k, sfk.view(B, s * sd))
        augv = aug.view(B, C, s, T // s, H, W)
        perm = torch.as_tensor(np.array([np.random.permutation(s) for _ in range(B)]),
                               dtype=torch.long, device=block.device)               # moco.py:544-546
        shuffled = torch.gather(augv, 2, perm.view(B, 1, s, 1, 1, 1).expand_as(augv)).reshape(B, C, T, H, W)
        dual = torch.cat([aug, shuffled], dim=0)
        _, dp = _run_modlist(self.encoder_q[:2], dual)
        dsf = F.normalize(self.series_proj_head_q(dp).view(2 * B, s, sd), dim=2)
        aug_sf, sh_sf = dsf[:B], dsf[B:]
        sh_sf = torch.scatter(sh_sf, 1, perm.view(B, s, 1).expand_as(sh_sf), sh_sf)
        ret.update(self.calc_ranking_loss(torch.stack([sf, sh_sf], dim=2), 2, 'unaug_ranking_', 0.5))
        ret.update(self.calc_ranking_loss(torch.stack([aug_sf, sh_sf], dim=2), 2, 'aug_ranking_', 0.5))
        return ret


class LinearClassifier(nn.Module):
    """Downstream classifier (model/classifier.py:10-70): backbone -> AdaptiveAvgPool3d(1) -> [l2 norm] -> [BatchNorm1d]
    -> [Dropout] -> Linear.  Same constructor arguments, sub-module names (`backbone`, `final_bn`, `final_fc`) and
    (logit, feature) return as the reference; `final_fc` weights N(0, 0.01), biases 0 (classifier.py:64-70)."""

    def __init__(self, num_class=101, network='resnet50', dropout=0.5, use_dropout=True, use_l2_norm=False,
                 use_final_bn=False, nonlinear=False, proj_dim=128):
        super().__init__()
        self.network, self.num_class, self.dropout = network, num_class, dropout
        self.use_dropout, self.use_l2_norm, self.use_final_bn = use_dropout, use_l2_norm, use_final_bn
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
        for name, prm in self.final_fc.named_parameters():
            if 'bias' in name:
                nn.init.constant_(prm, 0.0)
            elif 'weight' in name:
                nn.init.normal_(prm, mean=0.0, std=0.01)

    def forward(self, block):
        B = block.shape[0]
        feat3d = F.adaptive_avg_pool3d(self.backbone(block), (1, 1, 1)).view(B, self.param['feature_size'])
        if self.use_l2_norm:
            feat3d = F.normalize(feat3d, p=2, dim=1)
        logit = self.final_fc(self.final_bn(feat3d)) if self.use_final_bn else self.final_fc(feat3d)
        return logit, feat3d


def get_model(args):
    """pretrain.py:61-77."""
    kw = dict(n_series=args.n_series, series_dim=args.series_dim, series_T=args.series_T,
              aligned_T=args.aligned_T, mode=args.mode, args=args)
    if args.model == 'moco_naked':
        return MoCo_Naked(args.net, args.moco_dim, args.moco_k, args.moco_m, args.moco_t, args.distributed)
    if args.model == 'moco_timeseriesv4':
        return MoCo_TimeSeriesV4(args.net, args.moco_dim, args.moco_k, args.moco_m, args.moco_t,
                                 args.distributed, **kw)
    if args.model == 'simclr_naked':
        return SimCLR_Naked(args.net, args.moco_dim, args.moco_t, args.distributed)
    if args.model == 'simclr_timeseriesv4':
        return SimCLR_TimeSeriesV4(args.net, args.moco_dim, args.moco_t, args.distributed, **kw)
    raise NotImplementedError(args.model)


def train_step(model, block, optimizer):
    """The body of pretrain.py:394-451 without meters: forward, sum every '*loss', backward, step.
    Returns (ret dict, total loss)."""
    ret = model(block)
    loss = 0
    if 'clip_contrast_loss' in ret:
        loss = ret['clip_contrast_loss']
    for key in ret:
        if 'loss' in key and 'clip' not in key:
            loss = loss + ret[key]
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return ret, loss.detach()
