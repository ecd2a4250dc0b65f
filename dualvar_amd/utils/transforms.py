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


# ---------------------------------------------------------------------------------------------------------------
# Augmentation on the GPU.  The classes below keep the names, constructor arguments and -- call for
# call -- the RNG consumption (`random`, `numpy.random`) of the reference's tensor-side transforms
# (utils/transforms.py:201-373), but instead of touching pixels they fill one `dv_aug_frame` row per output frame;
# the pixels are produced by ONE pass of `dv_augment_ingest` straight into the stem's NDHWC input (FrameBatch below).
import math
import random

import numpy as np

AUG_NONE, AUG_BRIGHTNESS, AUG_CONTRAST, AUG_SATURATION, AUG_GRAY, AUG_HUE = 0, 1, 2, 3, 4, 5
AUG_MAX_OPS = 5
AUG_ROW = np.dtype([('src', '<i4'), ('crop_i', '<i4'), ('crop_j', '<i4'), ('crop_h', '<i4'), ('crop_w', '<i4'), ('flip', '<i4'),
                    ('op', '<i4', (AUG_MAX_OPS,)), ('factor', '<f4', (AUG_MAX_OPS,))])                   # include/dualvar_hip.h
AUG_BLUR = np.dtype([('radius', '<i4'), ('ww', '<u4'), ('fw', '<u4'), ('_pad', '<i4')])                  # dv_aug_blur


def box_blur_params(sigma, passes=3):
    """ImageFilter.GaussianBlur(radius=sigma) as Pillow runs it: `passes` extended box filters per axis (src/libImaging/
    BoxBlur.c).  -> (integer radius, ww, fw), the 8.24 fixed-point weights of the 2*radius+1 inner taps and of the two outer
    ones, computed with Pillow's own float32 sequence (`_gaussian_blur_radius`: Gwosdek et al., box length sqrt(12 s^2/n + 1))
    so that the integer kernel (csrc/augment.hip: aug_blur_kernel) reproduces PIL bit for bit."""
    f = np.float32
    radius = f(sigma)
    sigma2 = f(f(radius * radius) / f(passes))
    L = f(np.sqrt(12.0 * float(sigma2) + 1.0))
    lo = f(np.floor((float(L) - 1.0) / 2.0))
    a = f(f(f(2) * lo + f(1)) * f(f(lo * f(lo + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(lo + f(1)) * f(lo + f(1))))))
    fr = f(lo + a)
    if fr == 0:
        return 0, 0, 0
    r = int(fr)
    ww = int(f(16777216.0) / f(f(fr) * f(2) + f(1)))
    return r, ww, ((1 << 24) - (r * 2 + 1) * ww) // 2


class ClipState:
    """what the reference's transforms see as `vid` [C, N, h, w]: N frames with a common window, plus the ops so far"""

    def __init__(self, src, Hs, Ws):
        self.src = list(src)
        self.i, self.j, self.h, self.w = 0, 0, Hs, Ws          # window in the source frame
        self.out = None                                        # (H, W) once a Resize / RandomSizedCrop fixed it
        self.flip = False
        self.ops = []                                          # [(code, factors[N])], in applied order
        self.sigma = None                                      # per-frame Gaussian blur sigma (0: none), set by GaussianBlur

    @property
    def N(self):
        return len(self.src)

    def size(self):
        return self.out if self.out is not None else (self.h, self.w)

    def _no_colour_yet(self, what):
        if self.sigma is not None:
            raise ValueError('%s after the Gaussian blur: the blur is the last op of a dv_augment_ingest pipeline' % what)
        if self.ops:
            raise ValueError('%s after a colour op is not expressible in one dv_augment_ingest row' % what)

    def crop(self, i, j, h, w):
        self._no_colour_yet('crop')
        if self.out is not None:
            raise ValueError('crop after a resize is not expressible in one dv_augment_ingest row')
        if self.flip:
            raise ValueError('crop after a flip: put the crop first')
        self.i, self.j, self.h, self.w = self.i + i, self.j + j, h, w

    def rows(self, H, W):
        oh, ow = self.size()
        if (oh, ow) != (H, W):
            raise ValueError('pipeline produces %dx%d frames, the plan wants %dx%d' % (oh, ow, H, W))
        if len(self.ops) > AUG_MAX_OPS or sum(1 for c, _ in self.ops if c == AUG_CONTRAST) > 1:
            raise ValueError('at most %d colour ops and one contrast per frame' % AUG_MAX_OPS)
        t = np.zeros(self.N, dtype=AUG_ROW)
        t['src'], t['crop_i'], t['crop_j'], t['crop_h'], t['crop_w'] = self.src, self.i, self.j, self.h, self.w
        t['flip'] = int(self.flip)
        for n in range(self.N):
            k = 0
            for code, fac in self.ops:
                if code == AUG_GRAY and not fac[n]:
                    continue
                t['op'][n, k], t['factor'][n, k] = code, fac[n]
                k += 1
        return t

    def blur_rows(self):
        """dv_aug_blur rows of the clip's frames (all zero when the clip is not blurred)"""
        b = np.zeros(self.N, dtype=AUG_BLUR)
        if self.sigma is not None:
            for n in range(self.N):
                if self.sigma[n] > 0:
                    b['radius'][n], b['ww'][n], b['fw'][n] = box_blur_params(self.sigma[n])
        return b


def _corner(h, w, th, tw):
    """top-left corner of a th x tw window, rows first: the two `random.randint` draws of the reference's get_params"""
    top = random.randint(0, h - th)
    return top, random.randint(0, w - tw)


class RandomCrop(object):
    """transforms.py:201-219: no draw at all when the window already has the requested size"""

    def __init__(self, size):
        self.size = size

    @staticmethod
    def get_params(hw, output_size):
        (h, w), (th, tw) = hw, output_size
        if (h, w) == (th, tw):
            return 0, 0, h, w
        return _corner(h, w, th, tw) + (th, tw)

    def __call__(self, st):
        st.crop(*self.get_params(st.size(), self.size))
        return st


class RandomSizedCrop(object):
    """transforms.py:221-247: up to ten (area fraction in [0.5, 1], aspect in [3/4, 4/3]) proposals -- two uniform draws
    each, then the corner -- and a plain random crop of the output size when none fits; resized to `size`"""

    def __init__(self, size):
        self.size = size

    @staticmethod
    def get_params(hw, output_size):
        h, w = hw
        for _ in range(10):
            target = random.uniform(0.5, 1) * (h * w)
            aspect = random.uniform(3. / 4, 4. / 3)
            tw, th = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if th <= h and tw <= w:
                return _corner(h, w, th, tw) + (th, tw)
        th, tw = output_size
        return _corner(h, w, th, tw) + (th, tw)

    def __call__(self, st):
        st.crop(*self.get_params(st.size(), self.size))
        st.out = tuple(self.size)
        return st


class CenterCrop(object):                                   # transforms.py:17-24,250-255
    def __init__(self, size):
        self.size = size

    def __call__(self, st):
        h, w = st.size()
        th, tw = self.size
        st.crop(int(round((h - th) / 2.)), int(round((w - tw) / 2.)), th, tw)
        return st


class Resize(object):                                       # transforms.py:33-42,258-263 (size = (h, w))
    def __init__(self, size):
        if isinstance(size, int):
            raise NotImplementedError('Resize(int) rescales by a float factor; pass the (h, w) it produces')
        self.size = tuple(size)

    def __call__(self, st):
        st._no_colour_yet('resize')
        if st.out is not None:
            raise ValueError('two resizes in one pipeline')
        st.out = self.size
        return st


class RandomHorizontalFlip(object):                         # transforms.py:286-294
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, st):
        if random.random() < self.p:
            st._no_colour_yet('flip')
            st.flip = not st.flip
        return st


class RandomGray(object):                                   # transforms.py:80-88,305-311 (per-frame mask)
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, st):
        gray_map = np.random.uniform(size=(st.N,)) < self.p
        if gray_map.sum() == 0:
            return st
        st.ops.append((AUG_GRAY, gray_map.astype(np.float32)))
        return st


class ColorJitter(object):                                  # transforms.py:313-373
    """`hue` is an extension: the reference's tensor-side ColorJitter has none, its PIL one (utils/augmentation.py:429-508,
    the `ColorJitter(0.8, 0.8, 0.8, 0.2)` of pretrain.py:503) draws a shift in [-hue, hue] turns; here it joins the shuffled
    list like the other three and runs as DV_AUG_HUE (`adjust_hue_np`'s arithmetic)."""

    def __init__(self, brightness=0, contrast=0, saturation=0, consistent=False, p=1.0, n_channel=1, gray_channel=0, hue=0):
        self.brightness = self._check_input(brightness, 'brightness')
        self.contrast = self._check_input(contrast, 'contrast')
        self.saturation = self._check_input(saturation, 'saturation')
        self.hue = self._check_input(hue, 'hue', center=0, bound=(-0.5, 0.5))
        self.consistent, self.p = consistent, p

    @staticmethod
    def _check_input(value, name, center=1, bound=(0, float('inf'))):
        """a number v means [center - v, center + v]; a pair is taken as is; the identity range means "off" (None)"""
        if isinstance(value, (int, float)):
            if value < 0:
                raise ValueError('%s: a single number must not be negative' % name)
            lo, hi = center - value, center + value
        elif isinstance(value, (tuple, list)) and len(value) == 2:
            lo, hi = value
            if not bound[0] <= lo <= hi <= bound[1]:
                raise ValueError('%s range must lie in %s' % (name, (bound,)))
        else:
            raise TypeError('%s: a number or a (low, high) pair' % name)
        return None if lo == hi == center else [lo, hi]

    def _draw(self, rng, N):                                # random_adjust_*: transforms.py:165-190
        if self.consistent:
            return np.array([random.uniform(rng[0], rng[1])] * N)
        return np.random.uniform(rng[0], rng[1], size=(N,))

    def __call__(self, st):
        if random.random() < self.p:
            todo = []                                       # get_params: the list is shuffled BEFORE any factor is drawn
            if self.brightness is not None:
                todo.append((AUG_BRIGHTNESS, self.brightness))
            if self.contrast is not None:
                todo.append((AUG_CONTRAST, self.contrast))
            if self.saturation is not None:
                todo.append((AUG_SATURATION, self.saturation))
            if self.hue is not None:
                todo.append((AUG_HUE, self.hue))
            random.shuffle(todo)
            for code, rng in todo:
                st.ops.append((code, self._draw(rng, st.N).astype(np.float32)))
        return st


class GaussianBlur(object):
    """utils/augmentation.py:706-721 (the SimCLR blur, `A.GaussianBlur([.1, 2.], seq_len=...)` of pretrain.py:505): one sigma
    per block of `n_seqblock` frames (default: the whole clip), drawn with random.uniform as the reference draws it; the
    frames go ToPILImage -> PIL ImageFilter.GaussianBlur(radius=sigma) -> ToTensor.  Must be the last op of a pipeline: it
    works on the finished, re-quantised uint8 frame."""

    def __init__(self, sigma=(.1, 2.), seq_len=16, n_seqblock=0):
        self.sigma, self.seq_len = list(sigma), seq_len
        self.n_seqblock = n_seqblock if n_seqblock != 0 else seq_len

    def __call__(self, st):
        sig = np.zeros(st.N, dtype=np.float64)
        for idx in range(st.N):
            if idx % self.n_seqblock == 0:
                sigma = random.uniform(self.sigma[0], self.sigma[1])
            sig[idx] = sigma
        st.sigma = sig
        return st


class RandomApply(object):
    """torchvision.transforms.RandomApply as pretrain.py:499-505 uses it: the wrapped transforms run when a draw from
    torch's RNG is below p (`if self.p < torch.rand(1): return img`)"""

    def __init__(self, transforms, p=0.5):
        self.transforms, self.p = list(transforms), p

    def __call__(self, st):
        if self.p < float(torch.rand(1)):
            return st
        for t in self.transforms:
            st = t(st)
        return st


class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, st):
        for t in self.transforms:
            st = t(st)
        return st


class FrameBatch(object):
    """Decoded uint8 frames + one augmentation row per output frame: what the backbones' ingest consumes in place of a
    float clip tensor.  Quacks like the `[B, V, 3, T, H, W]` (or `[N, 3, T, H, W]`) tensor the models index."""

    def __init__(self, frames, table, shape, blur=None):
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError('frames must be uint8 [n_src, Hs, Ws, 3]')
        self.frames = frames.contiguous()
        self.table = table                                   # uint8 device tensor [rows * 64], rows in clip-major order
        self.blur = blur                                     # None, or uint8 device tensor [rows * 16] (dv_aug_blur), same order
        self.shape = torch.Size(shape)
        rows = 1
        for d in self.shape[:-4]:
            rows *= d
        if table.numel() != rows * self.shape[-3] * AUG_ROW.itemsize:
            raise ValueError('table has %d bytes, shape %s needs %d rows' % (table.numel(), tuple(shape), rows * self.shape[-3]))
        self.is_cuda, self.device, self.dtype, self.requires_grad = frames.is_cuda, frames.device, torch.float32, False

    @classmethod
    def build(cls, frames, clips, transform, size, views=1, device=None):
        """frames: uint8 [n_src, Hs, Ws, 3]; clips: per sample the source-frame indices of its T frames; every one of
        the `views` views of a sample draws its own augmentation (pretrain's two clips of a video)."""
        Hs, Ws = frames.shape[1:3]
        rows, blurs = [], []
        for src in clips:
            for _ in range(views):
                st = transform(ClipState(src, Hs, Ws))
                rows.append(st.rows(*size))
                blurs.append(st.blur_rows())
        T = len(clips[0])
        tab, btab = np.concatenate(rows), np.concatenate(blurs)
        device = device if device is not None else frames.device
        t = torch.from_numpy(tab.view(np.uint8).copy()).to(device)
        b = torch.from_numpy(btab.view(np.uint8).copy()).to(device) if btab['ww'].any() else None
        shape = (len(clips), views, 3, T) + tuple(size) if views > 1 else (len(clips), 3, T) + tuple(size)
        return cls(frames.to(device), t, shape, blur=b)

    def dim(self):
        return len(self.shape)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def contiguous(self):
        return self

    def float(self):
        return self

    def _flat(self):
        s = self.shape
        n = 1
        for d in s[:-4]:
            n *= d
        return FrameBatch(self.frames, self.table, (n,) + tuple(s[-4:]), blur=self.blur)

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        if len(shape) == 5 and shape[0] == -1 and tuple(shape[1:]) == tuple(self.shape[-4:]):
            return self._flat()
        if tuple(shape) == tuple(self.shape):
            return self
        raise NotImplementedError('FrameBatch only flattens its leading dimensions')

    view = reshape

    @staticmethod
    def cat(batches, dim=0):
        """torch.cat(..., dim=0) for frame batches over the SAME decoded frames: the augmentation tables are
        concatenated, no pixel is copied (MoCo_TimeSeriesV4 feeds [aug_x1 ; aug_x1] through one backbone pass,
        moco.py:551-556)."""
        if dim != 0:
            raise NotImplementedError('FrameBatch.cat joins the leading dimension only')
        first = batches[0]
        for b in batches[1:]:
            if b.frames.data_ptr() != first.frames.data_ptr() or tuple(b.shape[1:]) != tuple(first.shape[1:]):
                raise ValueError('FrameBatch.cat: batches must share their frames and per-sample shape')
        table = torch.cat([b.table.view(-1) for b in batches])
        blur = None
        if any(b.blur is not None for b in batches):
            blur = torch.cat([b.blur.view(-1) if b.blur is not None else
                              torch.zeros(b.table.numel() // AUG_ROW.itemsize * AUG_BLUR.itemsize, dtype=torch.uint8, device=b.table.device)
                              for b in batches])
        return FrameBatch(first.frames, table, (sum(b.shape[0] for b in batches),) + tuple(first.shape[1:]), blur=blur)

    def __getitem__(self, idx):
        # block[:, v]: one view of every sample
        if isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None) and isinstance(idx[1], int) and self.dim() == 6:
            B, V, _, T = self.shape[:4]
            rb = T * AUG_ROW.itemsize
            t = self.table.view(B, V, rb)[:, idx[1]].contiguous().view(-1)
            bl = None
            if self.blur is not None:
                bl = self.blur.view(B, V, T * AUG_BLUR.itemsize)[:, idx[1]].contiguous().view(-1)
            return FrameBatch(self.frames, t, (B,) + tuple(self.shape[2:]), blur=bl)
        raise NotImplementedError('FrameBatch supports block[:, v] only')
