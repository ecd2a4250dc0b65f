"""Shared test helpers (not the product)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CLIP = dict(T=8, H=112, W=112)


def gold(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.max(np.abs(got - ref)) / (np.max(np.abs(ref)) + 1e-12))


def grad_summary(model, P):
    sd = model.state_dict(keep_vars=True)
    out = {}
    for key in sorted(P.canonical_groups(model)):
        t = sd[key]
        if getattr(t, 'grad', None) is not None:
            g = t.grad.double()
            out[key] = np.array([g.abs().sum().item(), g.sum().item()])
    return out


def param_checksum(model, P):
    sd = model.state_dict()
    return {k: np.array([sd[k].double().abs().sum().item(), sd[k].double().sum().item()])
            for k in sorted(P.canonical_groups(model)) if sd[k].dtype.is_floating_point}


def total_loss(ret):
    """pretrain.py:402-445: clip loss + every other '*loss' entry."""
    loss = 0
    if 'clip_contrast_loss' in ret:
        loss = ret['clip_contrast_loss']
    for key in ret:
        if 'loss' in key and 'clip' not in key:
            loss = loss + ret[key]
    return loss
