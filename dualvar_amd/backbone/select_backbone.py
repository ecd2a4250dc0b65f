"""select_backbone(network, first_channel=3) -> (module, {'feature_size': int})
Same contract as the reference factory (backbone/select_backbone.py:7-31); the modules are the HIP-engine
backbones.  'r50' builds the 2D3D ResNet-50 the reference intended (its own call raises TypeError)."""
from .resnets import Bottleneck2d, Bottleneck3d, R2Plus1DNet, R3DNet, ResNet2d3d
from .s3dg import S3D


def select_backbone(network, first_channel=3):
    if network == 's3d':
        model = S3D(input_channel=first_channel)
    elif network == 's3dg':
        model = S3D(input_channel=first_channel, gating=True)
    elif network == 'r50':
        model = ResNet2d3d([Bottleneck2d, Bottleneck2d, Bottleneck3d, Bottleneck3d], [3, 4, 6, 3], first_channel)
    elif network == 'r21d':
        model = R2Plus1DNet()
    elif network == 'r3d':
        model = R3DNet()
    else:
        raise NotImplementedError(network)
    return model, {'feature_size': model.feature_size}
