"""utils/transforms.py:57-63,276-283 -- the only transform on the pretrain hot path.

`Normalize(mean, std, channel)` keeps the reference call signature; on the GPU path pretrain.py hands
its mean/std to the backbone (`set_input_normalization`) so the arithmetic is fused into the ingest
kernel and `__call__` becomes the identity for tensors tagged as fused."""
import torch


def normalize(vid, mean, std, channel=0):
    shape = [1] * vid.dim()
    shape[channel] = -1
    mean = torch.as_tensor(mean, dtype=vid.dtype, device=vid.device).view(shape)
    std = torch.as_tensor(std, dtype=vid.dtype, device=vid.device).view(shape)
    return (vid - mean) / std


class Normalize(object):
    def __init__(self, mean, std, channel=0, fused=False):
        self.mean, self.std, self.channel, self.fused = mean, std, channel, fused

    def __call__(self, vid):
        return vid if self.fused else normalize(vid, self.mean, self.std, self.channel)
