"""Host-side helpers with the reference's names (utils/utils.py): GatherLayer (the one custom operator of the
reference, :321-338), calc_topk_accuracy (:75-92), AverageMeter / ProgressMeter (:163-263), save_checkpoint
(:18-44), neq_load_customized (:112-137).  Device-agnostic: the collectives run on RCCL (backend 'nccl' on
ROCm) for GPU tensors and on gloo for the CPU tests."""
import glob
import os
from collections import deque

import torch
import torch.distributed as dist


class GatherLayer(torch.autograd.Function):
    """all_gather with autograd: forward returns the W per-rank tensors, backward keeps only this rank's
    gradient slice (no reduce-scatter) -- exactly utils/utils.py:321-338, which is what makes every rank's
    clip loss the *global* loss whose gradient DDP then averages."""

    @staticmethod
    def forward(ctx, x):
        world = dist.get_world_size()
        ctx.rank = dist.get_rank()
        x = x.contiguous()
        out = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(out, x)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        return grads[ctx.rank].clone()


def gather_features(x, distributed):
    """[B, ...] -> [N, ...] (N = B * world) through GatherLayer; identity when not distributed."""
    if not distributed:
        return x
    return torch.cat(GatherLayer.apply(x), dim=0)


@torch.no_grad()
def concat_all_gather(t):
    """moco.py:14-25 (no gradient)."""
    world = dist.get_world_size()
    t = t.contiguous()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0)


def calc_topk_accuracy(output, target, topk=(1,)):
    """utils/utils.py:75-92.  Kept for API compatibility (torch.topk); the train loop uses the fused
    `*_rank0` outputs of the loss kernels instead, which avoids the topk launch."""
    maxk = max(topk)
    n = target.size(0)
    pred = output.topk(maxk, 1, True, True)[1].t()
    hit = pred.eq(target.view(1, -1).expand_as(pred))
    return [hit[:k].reshape(-1).float().sum(0) * (1.0 / n) for k in topk]


def topk_from_rank(rank0, topk=(1,)):
    """hit@k == (number of negatives above the positive) < k."""
    return [(rank0 < k).float().mean() for k in topk]


class AverageMeter(object):
    """utils/utils.py:163-243 (value / running average / short local window)."""

    def __init__(self, name='null', fmt=':.4f'):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0
        self.local_history = deque([])
        self.local_avg = 0
        self.history = []

    def update(self, val, n=1, history=0, step=5):
        self.val = val
        self.sum += val * n
        self.count += n
        if n == 0:
            return
        self.avg = self.sum / self.count
        if history:
            self.history.append(val)
        if step > 0:
            self.local_history.append(val)
            if len(self.local_history) > step:
                self.local_history.popleft()
            self.local_avg = sum(self.local_history) / len(self.local_history)

    def __str__(self):
        fmtstr = '{name} {val' + self.fmt + '} ({avg' + self.fmt + '})'
        return fmtstr.format(**self.__dict__)


class ProgressMeter(object):
    def __init__(self, num_batches, meters, prefix='', logger=None):
        n = len(str(num_batches // 1))
        self.batch_fmtstr = '[{:' + str(n) + 'd}/' + ('{:' + str(n) + 'd}').format(num_batches) + ']'
        self.meters, self.prefix, self.logger = meters, prefix, logger

    def display(self, batch):
        line = '\t'.join([self.prefix + self.batch_fmtstr.format(batch)] + [str(m) for m in self.meters])
        (self.logger.info if self.logger is not None else print)(line)


def save_checkpoint(state, is_best=0, gap=1, filename='models/checkpoint.pth.tar', keep_all=False, is_save=True, top_k=5):
    """utils/utils.py:18-44: epochN.pth.tar + latest.pth.tar, at most `top_k` model_best_* files.
    Tensors are cloned out of the parameter arena so the file holds plain contiguous tensors."""
    if 'state_dict' in state:
        state = dict(state)
        state['state_dict'] = {k: v.detach().clone().contiguous().cpu() for k, v in state['state_dict'].items()}
    last = os.path.join(os.path.dirname(filename), 'epoch%s.pth.tar' % str(state['epoch'] - gap))
    if not keep_all and os.path.exists(last):
        os.remove(last)
    if is_save:
        torch.save(state, filename)
        torch.save(state, os.path.join(os.path.dirname(filename), 'latest.pth.tar'))
    if is_best:
        past = sorted(glob.glob(os.path.join(os.path.dirname(filename), 'model_best_*.pth.tar')),
                      key=lambda x: int(''.join(filter(str.isdigit, x))))
        while len(past) >= top_k:
            os.remove(past.pop(0))
        torch.save(state, os.path.join(os.path.dirname(filename), 'model_best_epoch%s.pth.tar' % str(state['epoch'])))


def neq_load_customized(model, pretrained_dict, verbose=True, args=None):
    """utils/utils.py:112-137: load the intersection of keys, report the rest."""
    log = args.logger.info if (args is not None and getattr(args, 'logger', None) is not None) else print
    own = model.state_dict()
    use = {k: v for k, v in pretrained_dict.items() if k in own}
    if verbose:
        log('weights not used from the file: %s' % [k for k in pretrained_dict if k not in own])
        log('weights not found in the file:   %s' % [k for k in own if k not in pretrained_dict])
    own.update(use)
    model.load_state_dict(own)
    return model
