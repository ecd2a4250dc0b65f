"""TEST INFRASTRUCTURE -- import shim for the *reference* (SURVEY.md Appendix B).

Only usable where /root/reference exists (the build container); never on the GPU box.
It stubs the third-party modules the reference imports but does not use on the hot
path (torchvision, IPython, numba, the proprietary `dataloader`), makes `.cuda()` the
identity so the CPU path runs (defect D6), and aliases the missing
`calc_contrast_loss` (defect D1).  Nothing from the reference is copied: it is
imported from where it lies and only its outputs are recorded (gen_golden.py).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('DUALVAR_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'model'))


def load_reference():
    """Returns a namespace with the reference's model / backbone modules."""
    if not available():
        raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    def stub(name, **attrs):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
        return sys.modules[name]

    tv = stub('torchvision')
    tv.transforms = stub('torchvision.transforms')
    stub('torchvision.utils')
    stub('torchvision.transforms.functional')
    stub('IPython', embed=lambda *a, **k: None)
    jit = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
    nb = stub('numba', jit=jit)
    nb.cuda = stub('numba.cuda', jit=jit)
    stub('dataloader', KVReader=object)

    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self                       # D6
    import model.simclr as S
    import model.moco as M
    import backbone.select_backbone as SB
    import backbone.resnet_2d3d as R
    import utils.utils as U
    import model.classifier as CL
    import utils.transforms as RT                                         # tensor-side transforms (functions need no torchvision)
    import utils.augmentation as RA                                       # PIL pipeline: only its numpy colour helpers are callable here
    S.SimCLR_TimeSeriesV4.calc_contrast_loss = S.SimCLR_TimeSeriesV4.calc_clip_contrast_loss   # D1
    M.MoCo_TimeSeriesV4.calc_contrast_loss = M.MoCo_TimeSeriesV4.calc_clip_contrast_loss       # D1

    def r50(first_channel=3):                                            # D7
        return R.ResNet2d3d([R.Bottleneck2d, R.Bottleneck2d, R.Bottleneck3d, R.Bottleneck3d],
                            [3, 4, 6, 3], input_channel=first_channel)

    def select_backbone(name, first_channel=3):
        if name == 'r50':
            return r50(first_channel), {'feature_size': 2048}
        return SB.select_backbone(name, first_channel)

    def linear_classifier(network, **kw):                                # D7 again: build around a working backbone factory
        CL.select_backbone = select_backbone
        return CL.LinearClassifier(network=network, **kw)

    return types.SimpleNamespace(simclr=S, moco=M, select_backbone=select_backbone, utils=U, r50=r50,
                                 linear_classifier=linear_classifier, transforms=RT, augmentation=RA)
