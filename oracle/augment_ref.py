"""TEST INFRASTRUCTURE -- CPU restatement of the augmenting ingest (include/dualvar_hip.h: dv_augment_ingest).

Follows the reference's tensor-side definitions in utils/transforms.py: crop :13-14, hflip :26-27, resize :33-42
(F.interpolate bilinear, align_corners=False), to_normalized_float_tensor :49-51, normalize :57-63, rgb_to_grayscale
:66-78, adjust_brightness / contrast / saturation :90-163 (`_blend` = clamp(ratio*a + (1-ratio)*b, 0, 1)).
Pinned against those functions themselves by oracle/gen_golden.py:case_augment (tests/golden/augment.npz).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

NONE, BRIGHTNESS, CONTRAST, SATURATION, GRAY, HUE = 0, 1, 2, 3, 4, 5
MAX_OPS = 5

# numpy mirror of `dv_aug_frame` (64 bytes per row)
ROW = np.dtype([('src', '<i4'), ('crop_i', '<i4'), ('crop_j', '<i4'), ('crop_h', '<i4'), ('crop_w', '<i4'), ('flip', '<i4'),
                ('op', '<i4', (MAX_OPS,)), ('factor', '<f4', (MAX_OPS,))])
assert ROW.itemsize == 64


def _luma(x):                                   # x [3, H, W]
    return 0.2989 * x[0] + 0.5870 * x[1] + 0.1140 * x[2]


def _blend(a, b, ratio):
    r = torch.tensor(float(ratio), dtype=torch.float32)
    return (r * a + (1 - r) * b).clamp(0, 1)


def _hue(x, factor):
    """utils/augmentation.py:26-106 (`_rgb2hsv_np`, `_hsv2rgb_np`, `adjust_hue_np`) on a [0, 1] float frame [3, H, W], without the
    uint8 re-quantisation of `adjust_hue_np` (the tensor-side pipeline stays in floats)"""
    r, g, b = x[0], x[1], x[2]
    maxc, minc = x.max(0).values, x.min(0).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / cd, (maxc - g) / cd, (maxc - b) / cd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    h = torch.remainder(h + torch.tensor(factor, dtype=torch.float32), 1.0)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int64) % 6
    v = maxc
    p = (v * (1.0 - s)).clamp(0, 1)
    q = (v * (1.0 - s * f)).clamp(0, 1)
    t = (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    pick = lambda opts: torch.stack(opts, 0).gather(0, i[None])[0]          # noqa: E731
    return torch.stack((pick([v, q, p, p, t, v]), pick([t, v, v, q, p, p]), pick([p, p, t, v, v, q])), 0)


def augment_frame(frames, row, H, W):
    """one output frame [3, H, W] fp32 in [0, 1] (before Normalize)"""
    fr = torch.from_numpy(np.ascontiguousarray(frames[int(row['src'])]))               # [Hs, Ws, 3] uint8
    x = fr.permute(2, 0, 1).to(torch.float32) / 255
    i, j, h, w = int(row['crop_i']), int(row['crop_j']), int(row['crop_h']), int(row['crop_w'])
    x = x[:, i:i + h, j:j + w]
    if (h, w) != (H, W):
        x = F.interpolate(x[None], size=(H, W), mode='bilinear', align_corners=False)[0]
    if int(row['flip']):
        x = x.flip(dims=(-1,))
    for k in range(MAX_OPS):
        op, f = int(row['op'][k]), float(row['factor'][k])
        if op == BRIGHTNESS:
            x = _blend(x, 0, f)
        elif op == CONTRAST:
            x = _blend(x, _luma(x).mean(-1).mean(-1), f)
        elif op == SATURATION:
            x = _blend(x, _luma(x)[None], f)
        elif op == GRAY:
            x = _luma(x)[None].expand(3, -1, -1)
        elif op == HUE:
            x = _hue(x, f)
    return x.contiguous()


def augment_ingest(frames, table, N, T, H, W, mean=None, std=None, perm=None):
    """frames uint8 [n_src, Hs, Ws, 3], table ROW[N*T] -> fp32 [N, 3, T, H, W] (the layout the backbones take)"""
    out = torch.empty(N, 3, T, H, W, dtype=torch.float32)
    for n in range(N):
        for t in range(T):
            ts = t
            if perm is not None:
                n_seg = perm.shape[1]
                seg = T // n_seg
                ts = int(perm[n, t // seg]) * seg + t % seg
            out[n, :, t] = augment_frame(frames, table[n * T + ts], H, W)
    if mean is not None:
        m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1, 1)
        s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1, 1)
        out = (out - m) / s
    return out
