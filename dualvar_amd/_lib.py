"""ctypes binding of libdualvar_hip.so (include/dualvar_hip.h).

There is deliberately NO fallback: if the shared library is missing or the device is not a
gfx950, every compute entry point raises.  The product never routes through PyTorch kernels
or the CPU oracle."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdualvar_hip.so')

ABI_VERSION = 2              # == DV_ABI_VERSION of include/dualvar_hip.h (tests/test_abi_and_host.py compares the two)
DV_F32, DV_BF16 = 0, 1
DV_BIAS, DV_RELU, DV_SIGMOID, DV_ACCUM, DV_STATS, DV_NO_RELU_MASK, DV_MASK_FROM_X = 1, 2, 4, 8, 16, 32, 64
DV_W3 = 128                  # conv fwd / dgrad in DV_F32: weights pre-split in fragment order (dv_pack_w3)

_ERR = {-1: 'DV_EINVAL (inconsistent shapes / unsupported parameter)',
        -2: 'DV_EALIGN (pointer or pitch misaligned)',
        -3: 'DV_EUNSUPPORTED'}


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'dtype', 'N', 'Ti', 'Hi', 'Wi', 'Cin', 'To', 'Ho', 'Wo', 'Cout', 'kt', 'kh', 'kw', 'st', 'sh', 'sw',
        'pt', 'ph', 'pw', 'cin_pitch', 'cout_pitch', 'ldx', 'ldy', 'flags')]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'dtype', 'N', 'Ti', 'Hi', 'Wi', 'C', 'To', 'Ho', 'Wo', 'kt', 'kh', 'kw', 'st', 'sh', 'sw',
        'pt', 'ph', 'pw', 'ldx', 'ldy')]


class PackDesc(C.Structure):
    _fields_ = [('src_off', C.c_int64), ('dst_off', C.c_int64), ('Cout', C.c_int32), ('Cin', C.c_int32),
                ('taps', C.c_int32), ('cin_pitch', C.c_int32), ('cout_pitch', C.c_int32), ('_pad', C.c_int32)]


class BnItem(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ('partials', 'local_stats', 'gamma', 'beta', 'running_mean', 'running_var', 'mean',
                                           'invstd', 'scale', 'shift', 'x', 'residual', 'y', 'dy', 'dx', 'dres', 'sums',
                                           'dgamma', 'dbeta')] +
                [('M', C.c_int64)] +
                [(n, C.c_int32) for n in ('n_tiles', 'tile_rows', 'pitch', 'C', 'ldx', 'ldr', 'ldy', 'lddy', 'lddx', 'lddres',
                                          'fwd_flags', 'bwd_flags', 'n_rep')] +
                [(n, C.c_float) for n in ('eps', 'momentum', 'inv_count', 'dparam_scale')] +
                [(n, C.c_int32) for n in ('blk_stats', 'blk_apply', 'blk_red', 'blk_bapply')] +
                [('red_ws', C.c_void_p)])


class BnReduce(C.Structure):
    """dv_bn_reduce: the BatchNorm in front of a conv, for dv_conv3d_dgrad_bn"""
    _fields_ = ([(n, C.c_void_p) for n in ('x', 'mean', 'invstd', 'scale', 'shift', 'sums')] +
                [(n, C.c_int32) for n in ('ldx', 'n_rep', 'flags', '_pad')])


class BnIn(C.Structure):
    """dv_bn_in: the BatchNorm (+ReLU) a conv applies to its input on load, for dv_conv3d_fwd_bn_in / dv_conv3d_wgrad_bn_in"""
    _fields_ = [('scale', C.c_void_p), ('shift', C.c_void_p), ('flags', C.c_int32), ('_pad', C.c_int32)]


class BnBwd(C.Structure):
    """dv_bn_bwd: the BatchNorm behind a conv whose input needs no gradient, for dv_conv3d_wgrad_bn"""
    _fields_ = ([(n, C.c_void_p) for n in ('x', 'mean', 'invstd', 'gamma', 'scale', 'shift', 'sums', 'dgamma', 'dbeta')] +
                [('inv_count', C.c_float), ('dparam_scale', C.c_float)] +
                [(n, C.c_int32) for n in ('ldx', 'n_rep', 'flags', '_pad')])


class W3Desc(C.Structure):
    _fields_ = [('src_off', C.c_int64), ('dst_off', C.c_int64), ('N', C.c_int32), ('Ktot', C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('bias', C.c_void_p),
                ('sam', C.c_int64), ('sak', C.c_int64), ('sbk', C.c_int64), ('sbn', C.c_int64), ('ldc', C.c_int64),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('flags', C.c_int32), ('tile_end', C.c_int32),
                ('alpha', C.c_float)]


P, I32, I64, F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
RETURNS_INT64 = ('dv_conv3d_wgrad_workspace', 'dv_conv3d_dgrad_bn_workspace', 'dv_w3_bytes', 'dv_bn_bwd_reduce_workspace', 'dv_infonce_workspace')       # everything else returns int
CD, PD = C.POINTER(ConvDesc), C.POINTER(PoolDesc)

# name -> argtypes, exactly as declared in include/dualvar_hip.h
SIGNATURES = {
    'dv_abi_version': [],
    'dv_check_device': [],
    'dv_conv3d_stat_tiles': [CD],
    'dv_conv3d_tile_rows': [CD],
    'dv_conv3d_tile_shape': [CD, I32, P, P],
    'dv_conv3d_ksplit_cols': [CD, I32],
    'dv_conv3d_tap_kind': [CD, I32],
    'dv_conv3d_tap_rows': [CD, I32],
    'dv_conv3d_fwd': [CD, P, P, P, P, P, P],
    'dv_conv3d_dgrad': [CD, P, P, P, P],
    'dv_conv3d_dgrad_bn': [CD, P, P, P, P, P],
    'dv_conv3d_dgrad_bn_workspace': [CD],
    'dv_conv3d_dgrad_bn_ws': [CD, P, P, P, P, P, I64, P],
    'dv_conv3d_wgrad_workspace': [CD],
    'dv_conv3d_wgrad_tile': [CD, P, P, P],
    'dv_conv3d_wgrad': [CD, P, P, P, P, I64, P],
    'dv_conv3d_wgrad_bn_ok': [CD],
    'dv_conv3d_bn_in_ok': [CD],
    'dv_conv3d_fwd_bn_in': [CD, P, P, P, P, P, P],
    'dv_conv3d_wgrad_bn_in': [CD, P, P, P, P, P, I64, P],
    'dv_conv3d_wgrad_bn': [CD, P, P, P, P, I64, P, P],
    'dv_quantize_fp8_workspace': [],
    'dv_quantize_fp8': [I32, P, I64, I32, I32, I32, P, I32, P, P, P],
    'dv_conv3d_fwd_fp8': [CD, P, P, P, P, P, P, P],
    'dv_conv3d_dgrad_fp8': [CD, P, P, P, P, P, P],
    'dv_w3_bytes': [I32, I32],
    'dv_pack_w3': [P, P, P, P, I32, P],
    'dv_pack_dgrad_weights': [I32, P, P, P, P, I32, P],
    'dv_cast_arena': [I32, P, P, I64, P],
    'dv_ingest_ncdhw': [I32, P, P, I32, I32, I32, I32, I32, I64, I32, P, P, P, I32, P],
    'dv_ingest_ncdhw_pad': [I32, P, P, I32, I32, I32, I32, I32, I64, I32, P, P, P, I32, I32, P],
    'dv_fill_cols_f32': [P, I64, I32, I32, I32, F, P],
    'dv_addcmul_f32': [P, P, P, F, I32, P],
    'dv_bn_eval_coeffs': [P, P, P, P, F, I32, P, P, P],
    'dv_bn_rows_partials_f32': [P, I32, I32, I32, P, P],
    'dv_softmax_ce_fwd': [P, I32, I32, I32, P, P, P, I32, P, P],
    'dv_knn_rank': [P, I32, I32, I32, P, P, P, P],
    'dv_softmax_rows_f32': [P, I32, I32, I32, P, I32, P],
    'dv_augment_ingest': [I32, P, I32, I32, I32, P, I32, I32, I32, I32, P, I32, I32, P, P, P, I32, P, P, P, P],
    'dv_bn_reduce_stats': [P, I32, I32, I32, I64, I32, P, P],
    'dv_bn_finalize': [P, I32, I32, I32, P, P, F, F, P, P, P, P, P, P, P],
    'dv_bn_stats_finalize': [P, I32, I32, I32, I64, I32, P, P, P, F, F, P, P, P, P, P, P, P],
    'dv_bn_apply': [I32, P, I32, P, P, P, I32, P, I32, I64, I32, I32, P],
    'dv_bn_stats_multi': [P, I32, I32, I32, P],
    'dv_bn_apply_multi': [I32, P, I32, I32, P],
    'dv_bn_finalize_multi': [P, I32, I32, P, P, I32, I32, P],
    'dv_bn_bwd_reduce_multi': [I32, P, I32, I32, P],
    'dv_bn_bwd_apply_multi': [I32, P, I32, I32, I32, P],
    'dv_bn_bwd_blocks': [I64, I32],
    'dv_bn_bwd_reduce': [I32, P, I32, P, I32, P, I32, P, P, I64, I32, I32, P, I32, P, P],
    'dv_bn_bwd_reduce_workspace': [I64, I32],
    'dv_bn_bwd_apply': [I32, P, I32, P, I32, P, I32, P, P, P, P, I32, F, F, P, P, P, I32, P, I32, I64, I32, I32, P],
    'dv_maxpool3d_fwd': [PD, P, P, P, P],
    'dv_maxpool3d_bwd': [PD, P, P, P, I32, P],
    'dv_bn_apply_maxpool': [PD, P, P, P, P, P, P],
    'dv_bn_bwd_reduce_maxpool': [PD, P, P, P, P, P, P, P, P, I32, P],
    'dv_bn_bwd_apply_maxpool': [PD, P, P, P, P, P, P, P, P, P, I32, F, F, P, P, P, I32, P],
    'dv_spatial_mean': [I32, P, I32, I32, I32, I32, P, P],
    'dv_spatial_mean_bwd': [I32, P, I32, I32, I32, P, I32, I32, P],
    'dv_gate_scale': [I32, P, I32, P, I32, I32, I32, P, I32, P],
    'dv_gate_bwd_reduce': [I32, P, I32, P, I32, P, I32, I32, I32, P, I32, P],
    'dv_gate_bwd_apply': [I32, P, I32, P, P, I32, I32, I32, P, I32, I32, P],
    'dv_colsum_f32': [P, I32, I32, I32, P, P],
    'dv_l2norm_fwd': [P, I32, I32, F, P, P, P],
    'dv_l2norm_bwd': [P, P, P, I32, I32, P, P],
    'dv_relu_bwd_f32': [P, P, I64, P, P],
    'dv_mean_f32': [P, I32, P, P],
    'dv_ntxent_fwd': [P, P, I32, I32, I32, I32, I32, F, P, P, P, P, P],
    'dv_infonce_fwd': [P, P, P, I32, I32, I32, F, P, P, P, P, P, P, I64, P],
    'dv_infonce_workspace': [I32, I32, I32],
    'dv_rank_margin': [P, I32, I32, I32, F, F, F, P, P, P, P, P],
    'dv_gemm_f32': [I32, I32, I32, P, I64, I64, P, I64, I64, P, I64, F, I32, P],
    'dv_gemm_f32_grouped': [P, I32, I32, P],
    'dv_gemm_f32_ex': [P, P],
    'dv_group_mean_f32': [P, I32, I32, I32, P, P],
    'dv_group_mean_bwd_f32': [P, I32, I32, I32, P, P],
    'dv_sgd_momentum': [P, P, P, I64, F, F, F, F, I32, P, P],
    'dv_ema': [P, P, I64, F, I32, P, P],
}

_lib = None


class DualVarHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DualVarHipError(
            f'{LIB_PATH} is missing: run `python -m dualvar_amd.build` (hipcc --offload-arch=gfx950). '
            'dualvar_amd has no CPU or PyTorch fallback.')
    # torch bundles its own libamdhip64; it must be in the process BEFORE our library so that both resolve to the
    # same HIP runtime (loading ours first binds it to /opt/rocm's copy and every launch then reports "no device")
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI and this table disagree
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name in RETURNS_INT64 else C.c_int
    got = lib.dv_abi_version()
    if got != ABI_VERSION:
        raise DualVarHipError(f'libdualvar_hip.so reports ABI version {got}, this binding was written against {ABI_VERSION}: '
                              'rebuild with `python -m dualvar_amd.build --force`')
    _lib = lib
    return lib


_device_ok = False


def require_device():
    """Fail loudly unless a gfx950 GPU is current."""
    global _device_ok
    if _device_ok:
        return
    import torch
    if not torch.cuda.is_available():
        raise DualVarHipError('dualvar_amd needs an MI355X (gfx950) GPU; none is visible and there is no fallback path')
    lib = load()
    arch = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), 'gcnArchName', '')
    if not arch.startswith('gfx950'):
        raise DualVarHipError('current device is %r, not gfx950; the kernels are built for MI355X only '
                              '(dv_check_device rc=%d)' % (arch, lib.dv_check_device()))
    _device_ok = True


def f32_exact():
    """DUALVAR_F32_EXACT as the LIBRARY reads it (csrc/conv.hip f32_exact(): atoi(value) != 0), so that the host's choice of
    pre-split weights and the kernels' choice of MFMA never disagree ('2', '01', ' 1' all mean on; 'yes' means off)."""
    import re
    m = re.match(r'\s*([+-]?\d+)', os.environ.get('DUALVAR_F32_EXACT', ''))
    return bool(m) and int(m.group(1)) != 0


# Workspaces of the ordered (ticketed) reductions -- dv_bn_bwd_reduce*, dv_infonce_fwd: their ticket words are "zero on entry,
# zero again on exit".  A launch that FAILS may leave tickets behind and would corrupt every later sum silently, so every such
# workspace registers here and check() re-zeroes them all after a failed launch.
import weakref as _weakref
_ticket_ws = []


def register_ticket_workspace(t):
    if len(_ticket_ws) >= 64 and len(_ticket_ws) % 64 == 0:     # plans come and go: drop the references of freed workspaces
        _ticket_ws[:] = [r for r in _ticket_ws if r() is not None]
    _ticket_ws.append(_weakref.ref(t))
    return t


def _rezero_ticket_workspaces():
    for r in list(_ticket_ws):
        t = r()
        if t is None:
            _ticket_ws.remove(r)
            continue
        try:
            t.zero_()
        except Exception:       # the device itself may be in an error state: nothing more to do here
            pass


def check(rc, name):
    if rc != 0:
        if rc > 0:              # a launch failed (hipError_t): ticket state unknown
            _rezero_ticket_workspaces()
        raise DualVarHipError(f'{name} failed: {_ERR.get(rc, "hipError_t %d" % rc)}')
