"""R(2+1)D, R3D and the 2D3D-ResNet-50 on the HIP engine.

References: backbone/r21d.py:11-266, backbone/r3d.py:10-157, backbone/resnet_2d3d.py:117-341.
Holders keep the reference's parameter names; residual adds are fused into the BatchNorm-apply kernel
(y = relu(bn(x) + shortcut)), so a block is conv/BN launches only."""
import math

import torch.nn as nn

from .base import HipBackbone, conv_geometry, emit_conv_bn, register_conv_bn


def _t3(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


class R21DConv(nn.Module):
    """(2+1)D factorisation: 1xkxk -> BN -> ReLU -> kx1x1; mid-channel formula r21d.py:47-49."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        k, s, p = _t3(kernel_size), _t3(stride), _t3(padding)
        mid = int(math.floor((k[0] * k[1] * k[2] * cin * cout) / (k[1] * k[2] * cin + k[0] * cout)))
        self.spatial_conv = nn.Conv3d(cin, mid, (1, k[1], k[2]), (1, s[1], s[2]), (0, p[1], p[2]), bias=False)
        self.bn = nn.BatchNorm3d(mid)
        self.temporal_conv = nn.Conv3d(mid, cout, (k[0], 1, 1), (s[0], 1, 1), (p[0], 0, 0), bias=False)

    def register(self, store, first=False):
        register_conv_bn(store, self.spatial_conv, self.bn, first)
        register_conv_bn(store, self.temporal_conv, None)

    def emit_raw(self, plan, x):
        """returns the un-normalised output of the temporal conv (its BN lives in the parent block)"""
        y = emit_conv_bn(plan, self.spatial_conv, self.bn, x)
        k, s, p = conv_geometry(self.temporal_conv)
        return plan.conv(plan.store.slot(self.temporal_conv.weight), y, k, s, p)


class R3DConv(nn.Module):
    """one dense 3-D conv (r3d.py:10-38)"""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        self.temporal_spatial_conv = nn.Conv3d(cin, cout, _t3(kernel_size), _t3(stride), _t3(padding), bias=False)

    def register(self, store, first=False):
        register_conv_bn(store, self.temporal_spatial_conv, None, first)

    def emit_raw(self, plan, x):
        k, s, p = conv_geometry(self.temporal_spatial_conv)
        return plan.conv(plan.store.slot(self.temporal_spatial_conv.weight), x, k, s, p)


class ResBlock(nn.Module):
    """conv-BN-ReLU-conv-BN (+ strided 1x1x1 shortcut) -> add -> ReLU  (r21d.py:73-122, r3d.py:41-89)"""

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

    def register(self, store):
        self.conv1.register(store); store.add_bn(self.bn1)
        self.conv2.register(store); store.add_bn(self.bn2)
        if self.downsample:
            self.downsampleconv.register(store); store.add_bn(self.downsamplebn)

    def emit(self, plan, x):
        res = plan.bn(self.bn1, self.conv1.emit_raw(plan, x), relu=True)
        raw2 = self.conv2.emit_raw(plan, res)
        shortcut = x
        if self.downsample:
            shortcut = plan.bn(self.downsamplebn, self.downsampleconv.emit_raw(plan, x), relu=False)
        return plan.bn(self.bn2, raw2, relu=True, residual=shortcut)


class ResLayer(nn.Module):
    def __init__(self, conv_t, cin, cout, kernel_size, layer_size, downsample=False):
        super().__init__()
        self.block1 = ResBlock(conv_t, cin, cout, kernel_size, downsample)
        self.blocks = nn.ModuleList([ResBlock(conv_t, cout, cout, kernel_size) for _ in range(layer_size - 1)])

    def register(self, store):
        self.block1.register(store)
        for b in self.blocks:
            b.register(store)

    def emit(self, plan, x):
        x = self.block1.emit(plan, x)
        for b in self.blocks:
            x = b.emit(plan, x)
        return x


class _ResNet18ish(HipBackbone):
    feature_size = 512
    conv_t = None

    def __init__(self, layer_sizes=(1, 1, 1, 1)):
        super().__init__()
        ct = self.conv_t
        self.conv1 = ct(3, 64, (3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3))
        self.bn1 = nn.BatchNorm3d(64)
        self.conv2 = ResLayer(ct, 64, 64, 3, layer_sizes[0])
        self.conv3 = ResLayer(ct, 64, 128, 3, layer_sizes[1], downsample=True)
        self.conv4 = ResLayer(ct, 128, 256, 3, layer_sizes[2], downsample=True)
        self.conv5 = ResLayer(ct, 256, 512, 3, layer_sizes[3], downsample=True)

    def register_params(self, store):
        self.conv1.register(store, first=True)
        store.add_bn(self.bn1)
        for layer in (self.conv2, self.conv3, self.conv4, self.conv5):
            layer.register(store)

    def emit(self, plan, x):
        x = plan.bn(self.bn1, self.conv1.emit_raw(plan, x), relu=True)
        for layer in (self.conv2, self.conv3, self.conv4, self.conv5):
            x = layer.emit(plan, x)
        return x


class R2Plus1DNet(_ResNet18ish):
    """r21d.py:214-266; (1,1,1,1) is the paper's 14.4 M-parameter net."""
    conv_t = R21DConv


class R3DNet(_ResNet18ish):
    """r3d.py:126-157"""
    conv_t = R3DConv


# ------------------------------------------------------------------ 2D3D ResNet-50
class _Bottleneck(nn.Module):
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

    def register(self, store):
        register_conv_bn(store, self.conv1, self.bn1)
        register_conv_bn(store, self.conv2, self.bn2)
        register_conv_bn(store, self.conv3, self.bn3)
        if self.downsample is not None:
            register_conv_bn(store, self.downsample[0], self.downsample[1])

    def emit(self, plan, x, force_relu):
        # conv1 (1x1x1 in the 2d blocks) and conv3 are the pointwise GEMMs BASELINE configs[4] puts on the fp8 matrix cores
        y = emit_conv_bn(plan, self.conv1, self.bn1, x, fp8=True)
        y = emit_conv_bn(plan, self.conv2, self.bn2, y)
        shortcut = x
        if self.downsample is not None:
            shortcut = emit_conv_bn(plan, self.downsample[0], self.downsample[1], x, relu=False)
        # the net ends with F.relu(x) (resnet_2d3d.py:341), so the final block's missing ReLU is applied here
        return emit_conv_bn(plan, self.conv3, self.bn3, y, relu=self.use_final_relu or force_relu, residual=shortcut, fp8=True)


class Bottleneck2d(_Bottleneck):
    temporal = False


class Bottleneck3d(_Bottleneck):
    temporal = True


class ResNet2d3d(HipBackbone):
    """resnet_2d3d.py:272-341 -- what `select_backbone('r50')` was meant to build."""
    feature_size = 2048

    def __init__(self, block, layers, input_channel=3):
        super().__init__()
        assert input_channel == 3
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
            downsample = nn.Sequential(nn.Conv3d(self.inplanes, planes * block.expansion, 1, cstride, bias=False),
                                       nn.BatchNorm3d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        n_plain = blocks - 2 if is_final else blocks - 1
        mods += [block(self.inplanes, planes) for _ in range(n_plain)]
        if is_final:
            mods.append(block(self.inplanes, planes, use_final_relu=False))
        return nn.Sequential(*mods)

    def register_params(self, store):
        register_conv_bn(store, self.conv1, self.bn1, first=True)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for b in layer:
                b.register(store)

    def emit(self, plan, x):
        x = emit_conv_bn(plan, self.conv1, self.bn1, x)
        mp = self.maxpool
        x = plan.maxpool(x, _t3(mp.kernel_size), _t3(mp.stride), _t3(mp.padding), sole_consumer=True)   # fused with bn1 + ReLU
        layers = (self.layer1, self.layer2, self.layer3, self.layer4)
        for li, layer in enumerate(layers):
            for bi, b in enumerate(layer):
                x = b.emit(plan, x, force_relu=(li == 3 and bi == len(layer) - 1))
        return x


# ------------------------------------------------------------------ 2D3D ResNet-18 (`r2d3d18`)
class BasicBlock2d(nn.Module):
    """resnet_2d3d.py:45-78: conv1x3x3-BN-ReLU-conv1x3x3-BN (+ identity / 1x1x1-conv shortcut) (+ReLU)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_final_relu=True):
        super().__init__()
        self.use_final_relu = use_final_relu
        self.conv1 = nn.Conv3d(inplanes, planes, (1, 3, 3), (1, stride, stride), (0, 1, 1), bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, (1, 3, 3), 1, (0, 1, 1), bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.downsample = downsample

    def register(self, store):
        register_conv_bn(store, self.conv1, self.bn1)
        register_conv_bn(store, self.conv2, self.bn2)
        if self.downsample is not None:
            register_conv_bn(store, self.downsample[0], self.downsample[1])

    def emit(self, plan, x):
        y = emit_conv_bn(plan, self.conv1, self.bn1, x)
        shortcut = x
        if self.downsample is not None:
            shortcut = emit_conv_bn(plan, self.downsample[0], self.downsample[1], x, relu=False)
        return emit_conv_bn(plan, self.conv2, self.bn2, y, relu=self.use_final_relu, residual=shortcut)


class ResNet2d3dFull(HipBackbone):
    """resnet_2d3d.py:203-270 `ResNet2d3d_full` as `r2d3d18()` builds it (:352-356): BasicBlock2d x [2,2,2,2]; the last
    block of layer4 (256 planes) has no final ReLU and `forward` adds none -- the feature map is NOT rectified."""
    feature_size = 256

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
        mods = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        n_plain = blocks - 2 if is_final else blocks - 1
        mods += [block(self.inplanes, planes) for _ in range(n_plain)]
        if is_final:
            mods.append(block(self.inplanes, planes, use_final_relu=False))
        return nn.Sequential(*mods)

    def register_params(self, store):
        register_conv_bn(store, self.conv1, self.bn1, first=True)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for b in layer:
                b.register(store)

    def emit(self, plan, x):
        x = emit_conv_bn(plan, self.conv1, self.bn1, x)
        mp = self.maxpool
        x = plan.maxpool(x, _t3(mp.kernel_size), _t3(mp.stride), _t3(mp.padding), sole_consumer=True)   # fused with bn1 + ReLU
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for b in layer:
                x = b.emit(plan, x)
        return x
