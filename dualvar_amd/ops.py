"""Thin Python launchers over the C ABI: shape checks on the host, raw pointers to the kernels.

`Act` is an NDHWC activation view (a channel slice of a [rows, ld] buffer).  Nothing here computes
with PyTorch; torch tensors are only device memory."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import DV_ACCUM, DV_BF16, DV_BIAS, DV_F32, DV_NO_RELU_MASK, DV_RELU, DV_SIGMOID, DV_STATS  # noqa: F401

TORCH_DTYPE = {DV_F32: torch.float32, DV_BF16: torch.bfloat16}
ESIZE = {DV_F32: 4, DV_BF16: 2}


def cp8(c):
    return (c + 7) & ~7


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class Act:
    """[N,T,H,W,C] view with row pitch `ld` starting at channel `off` of `buf` ([rows, ld_total])."""
    __slots__ = ('buf', 'N', 'T', 'H', 'W', 'C', 'ld', 'off', 'dtype', 'grad', 'cpitch', 'producer', 'hw_pad', 'bn_member')

    def __init__(self, buf, N, T, H, W, C, ld, off, dtype, cpitch=None):
        self.buf, self.N, self.T, self.H, self.W, self.C, self.ld, self.off, self.dtype = buf, N, T, H, W, C, ld, off, dtype
        self.cpitch = cpitch if cpitch is not None else cp8(C)
        self.grad = None
        self.producer = None
        self.hw_pad = 0              # zero border (pixels) materialised around H and W (the ingest frames)
        self.bn_member = None        # (BNGroupOp, BNMember) when this is the output of a BatchNorm (engine.Plan.bn_group)

    @property
    def rows(self):
        return self.N * self.T * self.H * self.W

    @property
    def S(self):
        return self.T * self.H * self.W

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.off * ESIZE[self.dtype]

    def slice(self, off, C_):
        assert off % 8 == 0 and off + cp8(C_) <= self.ld - self.off or off + C_ <= self.C
        return Act(self.buf, self.N, self.T, self.H, self.W, C_, self.ld, self.off + off, self.dtype)

    def like(self, device=None):
        return new_act(self.N, self.T, self.H, self.W, self.C, self.dtype, self.buf.device, cpitch=self.cpitch)


def new_act(N, T, H, W, C_, dtype, device, cpitch=None, zero=False):
    cpitch = cpitch if cpitch is not None else cp8(C_)
    rows = N * T * H * W
    alloc = torch.zeros if zero else torch.empty
    buf = alloc((rows, cpitch), dtype=TORCH_DTYPE[dtype], device=device)
    return Act(buf, N, T, H, W, C_, cpitch, 0, dtype, cpitch)


def _p(t):
    return 0 if t is None else (t.ptr if isinstance(t, Act) else t.data_ptr())


def conv_desc(dtype, x, y, k, s, p, flags=0):
    d = L.ConvDesc()
    d.dtype = dtype
    d.N, d.Ti, d.Hi, d.Wi, d.Cin = x.N, x.T, x.H, x.W, x.C
    d.To, d.Ho, d.Wo, d.Cout = y.T, y.H, y.W, y.C
    d.kt, d.kh, d.kw = k
    d.st, d.sh, d.sw = s
    d.pt, d.ph, d.pw = p
    d.cin_pitch, d.cout_pitch = x.cpitch, y.cpitch
    d.ldx, d.ldy = x.ld, y.ld
    d.flags = flags
    return d


def conv_out_dims(x, k, s, p):
    return tuple((i + 2 * pp - kk) // ss + 1 for i, kk, ss, pp in zip((x.T, x.H, x.W), k, s, p))


def conv_fwd(d, x, w, bias, y, stats):
    lib = L.load()
    L.check(lib.dv_conv3d_fwd(C.byref(d), _p(x), _p(w), _p(bias), _p(y), _p(stats), stream_ptr()), 'dv_conv3d_fwd')


def conv_dgrad(d, dy, wd, dx):
    lib = L.load()
    L.check(lib.dv_conv3d_dgrad(C.byref(d), _p(dy), _p(wd), _p(dx), stream_ptr()), 'dv_conv3d_dgrad')


def bn_reduce_desc(x, mean, invstd, scale, shift, sums, n_rep, flags=0):
    """dv_bn_reduce for dv_conv3d_dgrad_bn: the BatchNorm (input activation x, statistics, affine map of the forward) in front
    of the conv whose data gradient is being computed"""
    r = L.BnReduce()
    r.x, r.ldx = _p(x), x.ld
    r.mean, r.invstd, r.scale, r.shift, r.sums = (_p(t) for t in (mean, invstd, scale, shift, sums))
    r.n_rep, r.flags = n_rep, flags
    return r


def conv_dgrad_bn(d, dy, wd, dx, bn):
    lib = L.load()
    L.check(lib.dv_conv3d_dgrad_bn(C.byref(d), _p(dy), _p(wd), _p(dx), C.byref(bn), stream_ptr()), 'dv_conv3d_dgrad_bn')


def wgrad_workspace_bytes(d):
    return int(L.load().dv_conv3d_wgrad_workspace(C.byref(d)))


def conv_wgrad(d, x, dy, dw, workspace=None):
    """dw += x^T dy.  workspace: uint8/any tensor of >= wgrad_workspace_bytes(d) bytes (allocated here when omitted)"""
    lib = L.load()
    need = wgrad_workspace_bytes(d)
    if workspace is None and need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dw.device)
    nbytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    L.check(lib.dv_conv3d_wgrad(C.byref(d), _p(x), _p(dy), _p(dw), _p(workspace), nbytes, stream_ptr()), 'dv_conv3d_wgrad')


def bn_in_desc(scale, shift, relu=True):
    """dv_bn_in: the BatchNorm (+ReLU) a conv applies to its input on load (dv_conv3d_fwd_bn_in / dv_conv3d_wgrad_bn_in)"""
    r = L.BnIn()
    r.scale, r.shift, r.flags = _p(scale), _p(shift), (L.DV_RELU if relu else 0)
    return r


def conv_fwd_bn_in(d, x_bn, bn, w, y, stats):
    lib = L.load()
    L.check(lib.dv_conv3d_fwd_bn_in(C.byref(d), _p(x_bn), C.byref(bn), _p(w), _p(y), _p(stats), stream_ptr()), 'dv_conv3d_fwd_bn_in')


def conv_wgrad_bn_in(d, x_bn, bn, dy, dw, workspace=None):
    lib = L.load()
    need = wgrad_workspace_bytes(d)
    if workspace is None and need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dw.device)
    nbytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    L.check(lib.dv_conv3d_wgrad_bn_in(C.byref(d), _p(x_bn), C.byref(bn), _p(dy), _p(dw), _p(workspace), nbytes, stream_ptr()),
            'dv_conv3d_wgrad_bn_in')


def quantize_fp8(x, M, C, ld, fmt, dtype, q=None, scale=None, workspace=None):
    """per-tensor fp8 quantisation of the [M, C] view at `x` (pitch ld elements): -> (q uint8 [M, C], scale float32[1])"""
    lib = L.load()
    dev = x.device if isinstance(x, torch.Tensor) else x.buf.device
    q = torch.empty(M, C, dtype=torch.uint8, device=dev) if q is None else q
    scale = torch.empty(1, dtype=torch.float32, device=dev) if scale is None else scale
    if workspace is None:
        workspace = torch.empty(lib.dv_quantize_fp8_workspace() // 4, dtype=torch.float32, device=dev)
    L.check(lib.dv_quantize_fp8(dtype, _p(x), M, C, ld, fmt, _p(q), q.stride(0), _p(scale), _p(workspace), stream_ptr()),
            'dv_quantize_fp8')
    return q, scale


def conv_fwd_fp8(d, x8, w8, sx, sw, y, stats):
    L.check(L.load().dv_conv3d_fwd_fp8(C.byref(d), _p(x8), _p(w8), _p(sx), _p(sw), _p(y), _p(stats), stream_ptr()), 'dv_conv3d_fwd_fp8')


def conv_dgrad_fp8(d, dy8, wd8, sdy, sw, dx):
    L.check(L.load().dv_conv3d_dgrad_fp8(C.byref(d), _p(dy8), _p(wd8), _p(sdy), _p(sw), _p(dx), stream_ptr()), 'dv_conv3d_dgrad_fp8')


def pack_w3(w):
    """fp32 [rows, Ktot] (contiguous) -> the pre-split fragment-order copy dv_conv3d_fwd / dgrad take with DV_W3 (uint8 tensor)"""
    lib = L.load()
    rows, ktot = w.shape
    nbytes = int(lib.dv_w3_bytes(rows, ktot))
    out = torch.zeros(nbytes, dtype=torch.uint8, device=w.device)
    d = (L.W3Desc * 1)()
    d[0].src_off, d[0].dst_off, d[0].N, d[0].Ktot = 0, 0, rows, ktot
    descs = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(w.device)
    bmap = torch.tensor([(0, u) for u in range(0, nbytes // 48, 256)], dtype=torch.int32, device=w.device)
    call('dv_pack_w3', w.contiguous(), out, descs, bmap, bmap.shape[0])
    return out


def stat_tiles(d):
    return L.load().dv_conv3d_stat_tiles(C.byref(d))


def tile_rows(d):
    return L.load().dv_conv3d_tile_rows(C.byref(d))


def pool_desc(dtype, x, y, k, s, p):
    d = L.PoolDesc()
    d.dtype = dtype
    d.N, d.Ti, d.Hi, d.Wi, d.C = x.N, x.T, x.H, x.W, x.C
    d.To, d.Ho, d.Wo = y.T, y.H, y.W
    d.kt, d.kh, d.kw = k
    d.st, d.sh, d.sw = s
    d.pt, d.ph, d.pw = p
    d.ldx, d.ldy = x.ld, y.ld
    return d


def call(name, *args):
    """Generic launcher: converts Act / tensors to pointers and appends the current stream."""
    lib = L.load()
    conv = []
    for a in args:
        if isinstance(a, (Act, torch.Tensor)) or a is None:
            conv.append(_p(a))
        elif isinstance(a, C.Structure):
            conv.append(C.byref(a))
        else:
            conv.append(a)
    L.check(getattr(lib, name)(*conv, stream_ptr()), name)


# --------------------------------------------------------------------- layout helpers (host utilities)
def pack_weight(w, cin_pitch):
    """[O,I,kt,kh,kw] fp32 -> [O, taps, cin_pitch] fp32 (the master / forward layout)."""
    O, I = w.shape[:2]
    taps = w[0, 0].numel()
    out = torch.zeros(O, taps, cin_pitch, dtype=torch.float32, device=w.device)
    out[:, :, :I] = w.reshape(O, I, taps).permute(0, 2, 1)
    return out


def unpack_weight(wp, shape):
    O, I = shape[:2]
    taps = wp.shape[1]
    return wp[:, :, :I].permute(0, 2, 1).reshape(shape).contiguous()


def act_from_ncdhw(x, dtype, cpitch=None):
    """Test/host utility: NCDHW float tensor -> Act (uses torch ops; not on the product hot path)."""
    N, C_, T, H, W = x.shape
    a = new_act(N, T, H, W, C_, dtype, x.device, cpitch=cpitch, zero=True)
    a.buf[:, :C_] = x.permute(0, 2, 3, 4, 1).reshape(-1, C_).to(TORCH_DTYPE[dtype])
    return a


def act_to_ncdhw(a):
    v = a.buf[:, a.off:a.off + a.C].float().reshape(a.N, a.T, a.H, a.W, a.C)
    return v.permute(0, 4, 1, 2, 3).contiguous()
