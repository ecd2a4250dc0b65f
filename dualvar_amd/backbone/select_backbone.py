"""`select_backbone(network, first_channel=3) -> (module, {'feature_size': int})`.

Contract of the reference factory (backbone/select_backbone.py:7-31): the returned module maps clips
[N,3,T,H,W] to a post-ReLU feature map [N,feature_size,T',H',W'].  Here the modules are the HIP-engine
backbones, and 'r50' builds the 2D3D ResNet-50 the reference meant to build (its own call raises TypeError:
SURVEY.md D7).  Every name of the reference factory is provided (s3d, s3dg, r21d, r3d, r50, r2d3d18, c3d)."""
from .c3d import C3D
from .resnets import Bottleneck2d, Bottleneck3d, R2Plus1DNet, R3DNet, ResNet2d3d, ResNet2d3dFull
from .s3dg import S3D

_FACTORIES = {
    's3d': lambda ch: S3D(input_channel=ch, gating=False),
    's3dg': lambda ch: S3D(input_channel=ch, gating=True),
    'r21d': lambda ch: R2Plus1DNet(),
    'r3d': lambda ch: R3DNet(),
    'r50': lambda ch: ResNet2d3d([Bottleneck2d, Bottleneck2d, Bottleneck3d, Bottleneck3d], [3, 4, 6, 3], ch),
    'r2d3d18': lambda ch: ResNet2d3dFull(),
    'c3d': lambda ch: C3D(),
}


def select_backbone(network, first_channel=3):
    try:
        module = _FACTORIES[network](first_channel)
    except KeyError:
        raise NotImplementedError('backbone %r is not part of the MI355X pretrain path (have: %s)'
                                  % (network, ', '.join(sorted(_FACTORIES)))) from None
    return module, {'feature_size': module.feature_size}
