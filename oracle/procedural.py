"""TEST INFRASTRUCTURE -- RNG-independent fills shared by the fixture generator, oracle and tests.

Weights are never shipped (46-60 MB); instead every float tensor of a state_dict is
filled as a deterministic function (numpy legacy RandomState stream) of its *canonical key* (the lexicographically
smallest of its alias names -- S3D registers the stem twice) and its shape, so the
reference, the oracle and the HIP product can all be initialised identically without
depending on any torch RNG stream.
"""
import zlib

import numpy as np
import torch

PHI = 0.6180339887498949


def _wave(numel, seed, freq=1.0):
    """Deterministic pseudo-noise in [-1, 1]: a golden-ratio phase walk pushed through sin."""
    i = np.arange(numel, dtype=np.float64)
    phase = (seed % 9973) * 0.7548776662466927
    return np.sin((i * (PHI * 7.0 + freq) + phase) * 2.3999632297286533 + 0.37 * np.sin(i * 0.01 + phase))


def canonical_groups(module):
    """{canonical_key: [alias keys]} over state_dict tensors that share storage."""
    groups = {}
    for k, v in module.state_dict(keep_vars=True).items():
        groups.setdefault((v.data_ptr(), tuple(v.shape), v.dtype), []).append(k)
    return {min(ks): sorted(ks) for ks in groups.values()}


@torch.no_grad()
def procedural_init(module, gain=1.0):
    """Fill every parameter / float buffer of `module` from its canonical key.

    Values are iid Gaussian draws from numpy's legacy RandomState(crc32(key)) -- a generator whose
    stream is frozen by numpy's compatibility policy, so the fill does not depend on the torch version.
    (A smooth closed-form fill was tried first: it makes S3D-G's BatchNorm channels nearly degenerate and
    the reference's own fp32 output then differs from its fp64 output by 10 %, useless as a parity pin.)

    conv / linear weights : N(0, 1/fan_in)          biases              : 0.1 * N(0,1)
    BN weight / bias      : 1 + 0.1*N / 0.1*N       running_mean / var  : 0 / 1      counters : 0
    MoCo queues           : N(0,1), then column-normalised like moco.py:79-81,318-323
    """
    sd = module.state_dict(keep_vars=True)
    for key in sorted(canonical_groups(module)):
        t = sd[key]
        seed = zlib.crc32(key.encode()) & 0x7FFFFFFF
        leaf = key.rsplit('.', 1)[-1]
        if not t.dtype.is_floating_point:
            t.zero_()
            continue
        n = t.numel()
        draw = torch.from_numpy(np.random.RandomState(seed).standard_normal(n)).to(t.dtype)
        if leaf == 'running_mean':
            t.zero_()
        elif leaf == 'running_var':
            t.fill_(1.0)
        elif leaf in ('queue', 'series_queue'):
            t.copy_(draw.reshape(t.shape))
        elif t.dim() == 1:
            if leaf == 'weight':                      # BN gamma
                t.copy_(1.0 + 0.1 * draw)
            else:                                     # BN beta, conv / linear bias
                t.copy_(0.1 * draw)
        else:
            fan_in = n // t.shape[0]
            t.copy_((gain * (1.0 / fan_in) ** 0.5) * draw.reshape(t.shape))
    # derived buffers (normalised queues)
    if 'queue' in sd:
        q = sd['queue']
        q.copy_(torch.nn.functional.normalize(q, dim=0))
    if 'series_queue' in sd:
        sq = sd['series_queue']
        K = sq.shape[1]
        s = getattr(module, 'n_series', 2)
        sq.copy_(torch.nn.functional.normalize(sq.view(s, -1, K), dim=1).view(-1, K))
    # MoCo: key encoder starts as a copy of the query encoder (moco.py:74-76,310-315)
    for key, t in sd.items():
        for kpre, qpre in (('encoder_k.', 'encoder_q.'), ('series_proj_head_k.', 'series_proj_head_q.')):
            if key.startswith(kpre) and t.dtype.is_floating_point:
                t.copy_(sd[qpre + key[len(kpre):]])
    return module


def procedural_clips(B, V, T, H, W, seed=1234, C=3):
    """Synthetic already-normalised clip block [B, V, C, T, H, W].

    Every sample gets its own spatial / temporal frequencies, amplitude and per-channel offset
    (views of one sample share them up to a phase), plus low-amplitude pseudo-noise, so that the
    pooled features of different samples differ and the contrastive logits are not degenerate."""
    out = np.zeros((B, V, C, T, H, W))
    t = np.arange(T)[:, None, None]
    h = np.arange(H)[None, :, None]
    w = np.arange(W)[None, None, :]
    for b in range(B):
        r = _wave(8, seed * 31 + b * 7 + 1)
        fx, fy, ft = 0.08 + 0.9 * abs(r[0]), 0.08 + 0.9 * abs(r[1]), 0.2 + 1.2 * abs(r[2])
        for v in range(V):
            for c in range(C):
                ph = 2.1 * c + 0.9 * v + 3 * r[3]
                base = np.sin(fx * w + fy * h * (1 + 0.2 * v) + ft * t + ph) * (0.6 + 0.8 * abs(r[4]))
                base = base + 0.5 * r[5 + (c % 3)] + 0.4 * np.sign(np.sin(0.5 * fx * w * h / (1 + b) + ph))
                out[b, v, c] = base
    out = out + 0.35 * _wave(out.size, seed, freq=0.31).reshape(out.shape)
    return torch.from_numpy(out).float()


def procedural_unit_features(*shape, seed=7):
    n = int(np.prod(shape))
    x = torch.from_numpy(_wave(n, seed, freq=0.13)).float().reshape(*shape)
    return torch.nn.functional.normalize(x, dim=-1)


# ---------------------------------------------------------------------------------------------------------------
# element-wise gradient pins (a permutation-invariant checksum would pass a transposed tap / channel)
GSAMPLE_FAMILIES = ('gating', 'fc', 'Conv_1a', 'conv1.', 'Conv_2b', 'Conv_2c', 'branch0.0', 'branch1.1.conv1', 'branch1.1.conv2',
                    'branch2.1.conv1', 'branch3.1', 'downsample', 'encoder_q.2', 'encoder_q.4', 'series_proj_head')


def grad_sample_keys(module, n_even=14):
    """Canonical keys of the tensors whose gradients are stored element-wise: the first tensor of every op family of
    GSAMPLE_FAMILIES (stem convs, merged branch-entry 1x1x1, separable pairs, the conv after the 3x3x3 pool, self-gating
    fc, heads, ...) with its BatchNorm neighbours, plus `n_even` tensors spread evenly over the rest."""
    sd = module.state_dict(keep_vars=True)
    keys = [k for k in sorted(canonical_groups(module)) if getattr(sd[k], 'requires_grad', False)]
    pick = []
    for fam in GSAMPLE_FAMILIES:
        hit = [k for k in keys if fam in k]
        pick += hit[:2]
    step = max(1, len(keys) // n_even)
    pick += keys[::step]
    seen, out = set(), []
    for k in pick:
        if k not in seen:
            seen.add(k)
            out.append(k)
    return out


def grad_sample_index(numel):
    """the sampled flat indices of a tensor with `numel` elements: every 97th, thinned to <= ~1000 samples"""
    stride = 97
    while numel // stride > 1000:
        stride += 194                    # stays odd: no aliasing with the (even) channel / tap pitches
    return np.arange(0, numel, stride)


def grad_samples(module):
    """{canonical key: float64 samples of .grad (flattened in the parameter's own [O, I, kt, kh, kw] order)}"""
    sd = module.state_dict(keep_vars=True)
    out = {}
    for k in grad_sample_keys(module):
        g = sd[k].grad
        if g is None:
            continue
        flat = g.detach().double().cpu().reshape(-1).numpy()
        out[k] = flat[grad_sample_index(flat.size)].copy()
    return out
