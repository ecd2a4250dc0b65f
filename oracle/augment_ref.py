"""TEST INFRASTRUCTURE -- CPU restatement of the augmenting ingest (include/dualvar_hip.h: dv_augment_ingest).

Follows the reference's tensor-side definitions in utils/transforms.py: crop :13-14, hflip :26-27, resize :33-42
(F.interpolate bilinear, align_corners=False), to_normalized_float_tensor :49-51, normalize :57-63, rgb_to_grayscale
:66-78, adjust_brightness / contrast / saturation :90-163 (`_blend` = clamp(ratio*a + (1-ratio)*b, 0, 1)).
Gaussian blur: the reference applies PIL's ImageFilter.GaussianBlur to the uint8 frame (utils/augmentation.py:706-721).  PIL
(Pillow -- a third-party dependency the reference does not pin; 12.2.0 in this image) is absent from /root/reference, so its
published algorithm is restated here (src/libImaging/BoxBlur.c: three passes per axis of an extended box filter in 8.24 fixed
point, radius from sigma by Gwosdek et al.'s formula in float32) and pinned bit for bit against PIL itself by
oracle/gen_golden.py:case_augment (tests/golden/augment.npz part D).
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


# numpy mirror of `dv_aug_blur` (16 bytes per row)
BLUR = np.dtype([('radius', '<i4'), ('ww', '<u4'), ('fw', '<u4'), ('_pad', '<i4')])


def box_blur_params(sigma, passes=3):
    """(int radius, ww, fw) of Pillow's extended box filter for ImageFilter.GaussianBlur(radius=sigma): BoxBlur.c
    `_gaussian_blur_radius` (all `float` variables, the sqrt and the floor in double) and ImagingHorizontalBoxBlur's weights"""
    f = np.float32
    radius = f(sigma)
    sigma2 = f(f(radius * radius) / f(passes))
    L = f(np.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(np.floor((float(L) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    fr = f(l + a)
    if fr == 0:
        return 0, 0, 0
    r = int(fr)
    ww = int(f(16777216.0) / f(f(fr) * f(2) + f(1)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def _box_line_pass(img, r, ww, fw):
    n = img.shape[-1]
    x = np.arange(n)
    a = img.astype(np.int64)
    acc = np.zeros(img.shape, dtype=np.int64)
    for d in range(-r, r + 1):
        acc += a[..., np.clip(x + d, 0, n - 1)]
    far = a[..., np.clip(x - r - 1, 0, n - 1)] + a[..., np.clip(x + r + 1, 0, n - 1)]
    bulk = (acc * ww + far * fw) & 0xFFFFFFFF
    return ((bulk + (1 << 23)) >> 24).astype(np.uint8)


def gaussian_blur_u8(u8, r, ww, fw, passes=3):
    """u8 [H, W, C] -> PIL's GaussianBlur of it: `passes` horizontal then `passes` vertical box passes, each re-quantised"""
    if ww == 0:
        return u8
    o = np.moveaxis(u8, 1, -1)
    for _ in range(passes):
        o = _box_line_pass(o, r, ww, fw)
    out = np.moveaxis(o, -1, 1)
    o = np.moveaxis(out, 0, -1)
    for _ in range(passes):
        o = _box_line_pass(o, r, ww, fw)
    return np.ascontiguousarray(np.moveaxis(o, -1, 0))


def blur_frame(x, brow):
    """x [3, H, W] float in [0, 1] -> ToPILImage (mul(255).byte()) -> blur -> ToTensor (/255)"""
    u8 = (x * 255).to(torch.uint8).permute(1, 2, 0).numpy()
    out = gaussian_blur_u8(u8, int(brow['radius']), int(brow['ww']), int(brow['fw']))
    return torch.from_numpy(out).permute(2, 0, 1).to(torch.float32) / 255


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


def augment_ingest(frames, table, N, T, H, W, mean=None, std=None, perm=None, blur=None):
    """frames uint8 [n_src, Hs, Ws, 3], table ROW[N*T] (+ optional BLUR[N*T]) -> fp32 [N, 3, T, H, W] (the layout the
    backbones take)"""
    out = torch.empty(N, 3, T, H, W, dtype=torch.float32)
    for n in range(N):
        for t in range(T):
            ts = t
            if perm is not None:
                n_seg = perm.shape[1]
                seg = T // n_seg
                ts = int(perm[n, t // seg]) * seg + t % seg
            x = augment_frame(frames, table[n * T + ts], H, W)
            if blur is not None and int(blur[n * T + ts]['ww']) != 0:
                x = blur_frame(x, blur[n * T + ts])
            out[n, :, t] = x
    if mean is not None:
        m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1, 1)
        s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1, 1)
        out = (out - m) / s
    return out
