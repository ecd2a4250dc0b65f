"""Execution engine: flat parameter arenas + static launch plans over the HIP kernels.

A backbone is described once as a graph of ops on NDHWC activations (`Plan`); the plan owns
every buffer (activations, their gradients, BatchNorm scratch) and replays a fixed list of kernel
launches for forward and for backward.  Nothing is traced or compiled: the lists are built by
ordinary Python that mirrors the reference's module structure, and they are replayed on the
current HIP stream (so a whole step can be captured into a hipGraph).

`ParamStore` keeps the fp32 master weights of one encoder in a single arena laid out
[Cout][taps][Cin_pad] (K-contiguous, what the implicit-GEMM kernels read), with the nn.Parameters
re-pointed to strided views of it so that state_dict()/optimizers see the reference's shapes.
Gradients accumulate in a twin arena (one flat all-reduce for data parallel), bf16 compute
copies and the dgrad-layout weights are refreshed by two launches after every update.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import rccl
from . import ops
from ._lib import DV_MASK_FROM_X
from .ops import DV_ACCUM, DV_BF16, DV_BIAS, DV_F32, DV_NO_RELU_MASK, DV_RELU, DV_SIGMOID, DV_STATS, Act, cp8


BN_REPLICAS = 1       # the BN-backward sums are reduced in block order (dv_bn_bwd_reduce's ordered form): one final copy


def _align8(n):
    return (n + 7) & ~7


# weight gradients on a side stream (see Plan.run_backward); DUALVAR_WGRAD_STREAM=0 keeps everything on one stream
WGRAD_SIDE_STREAM = os.environ.get('DUALVAR_WGRAD_STREAM', '1') != '0'
# backward launches whose results only the optimizer reads: weight gradients, the gate FCs' bias gradients
SIDE_LAUNCHES = ('conv_wgrad', 'gate_db')
# side launches issued per main-stream event (the event marker costs the main stream a few microseconds each)
WGRAD_BATCH = int(os.environ.get('DUALVAR_WGRAD_BATCH', '4'))
# ... but a LARGE weight gradient (>= WGRAD_FLUSH_FLOPS) is released at once and runs beside its own layer's data gradient --
# where that pays.  Batched, the last ones of an S3D-G pass (the stem's and Conv_2c's, 0.4 - 0.6 ms each) waited for the end of the
# main chain although their operands had been ready for up to a millisecond (tools/step_timeline.py: 1.6 ms of the step ran
# nothing but those; 0.65 ms now), step 16.41 -> 16.15 ms.  It pays where the main chain has bandwidth-bound work for the weight
# gradients to run beside (BatchNorm backward, pools, gating); where the chain is as matrix-bound as they are, two such kernels
# on one GPU are slower than one after the other: R3D 26.5 -> 27.6 ms, even for its last four only.  The plan decides from its own
# composition: estimated time of the chain's non-conv launches / of its data gradients (S3D-G 1.03, r50 0.93, R(2+1)D 0.65 --
# measured gain, gain, neutral --, R3D 0.27 -- measured loss): early release at >= WGRAD_EARLY_RATIO.
WGRAD_FLUSH_FLOPS = float(os.environ.get('DUALVAR_WGRAD_FLUSH_GFLOP', '3')) * 1e9
WGRAD_EARLY_RATIO = 0.5
WGRAD_EARLY = os.environ.get('DUALVAR_WGRAD_EARLY', 'auto')        # 'auto' | '0' | '1' (A/B)
# BatchNorm + ReLU whose only consumer is a max-pool (the stems) run fused with it (module switch: the tests compare both forms)
FUSE_BN_POOL = True
# conv -> BatchNorm -> conv with a single reader: the BatchNorm-backward reduce can run in the second conv's data-gradient
# epilogue (dv_conv3d_dgrad_bn).  OFF by default -- measured on the S3D-G step (MI355X, fp32 / bf16): the reduce launches shrink
# 1.36 -> 0.66 / 1.07 -> 0.55 ms per step, but the data gradients that carry them grow by 0.62 / 0.52 ms (a serial tail per
# workgroup that re-reads its tile and the BatchNorm input), and they sit on the critical path: 20.47 -> 20.77 / 9.79 -> 9.98 ms.
FUSE_BN_REDUCE = False        # (module switch, no environment variable: tests/test_models_gpu.py monkeypatches it)
# The same fusion in its ORDERED form on the LDS-staged input-tile kernel (dv_conv3d_dgrad_bn_ws: the sums come from the
# accumulators and one read of the BatchNorm's input, tile rows are folded in tile order -- no float atomics): taken wherever the
# consuming conv's data gradient runs on that kernel (fp32 mode: the separable pairs and the strided stem conv of S3D-G / R(2+1)D).
# OFF by default since the end of round 4: while the step's tail was weight gradients waiting for the end of the main chain it was
# a wash (data gradients +0.35 ms, reduce launches -0.25 ms, step equal); with the large weight gradients released early
# (WGRAD_EARLY_RATIO) the main chain is the critical path again and the fusion costs 0.15 ms (16.23 vs 16.38 ms, three A/B pairs;
# every K threshold between 512 and 2 900 lies in between).
FUSE_BN_REDUCE_TAP = os.environ.get('DUALVAR_FUSE_BN_REDUCE_TAP', '0') != '0'      # (A/B switch; tests monkeypatch the module attribute)
# ... only behind a long K loop (>= 512 = taps x channel pitch of dY) and never for the strided stem conv: the epilogue's read of
# the BatchNorm input is exposed at the end of a workgroup's life, and for a short loop it costs more than the standalone reduce
# saves -- all eight candidates of the S3D-G step fused: data gradients +740 us, reduce launches -565 us (the stem conv alone 422 ->
# 854 us for a 312 us reduce); with this rule the step is 0.03 - 0.05 ms faster, i.e. the fusion VERDICT round 3 priced at -0.26 ms
# is worth a tenth of that.
FUSE_BN_REDUCE_TAP_MIN_K = int(os.environ.get('DUALVAR_FUSE_BN_REDUCE_TAP_MIN_K', '512'))
# BatchNorm-backward APPLY inside the weight gradient of a conv whose input needs no gradient (the first conv of a network:
# dL/d(conv output) has that weight gradient as its only reader): dv_conv3d_wgrad_bn forms it on the fly from dL/dy and the
# conv output -- one read of each instead of read + read + write (apply) + read (wgrad).  fp32 split mode only.
FUSE_BN_WGRAD = os.environ.get('DUALVAR_FUSE_BN_WGRAD', '1') != '0'      # (A/B switch; tests monkeypatch the module attribute)
# BatchNorm ON LOAD (include/dualvar_hip.h: dv_conv3d_fwd_bn_in / dv_conv3d_wgrad_bn_in): conv -> BN -> ReLU -> conv chains whose
# second conv is the only reader of the BatchNorm's output (the 1xkxk -> kx1x1 pairs of backbone/s3dg.py:30-65 and the stem): the
# output is never written; the second conv's forward and weight gradient read the BatchNorm's INPUT and apply the affine map +
# ReLU while they stage it.  Priced before it was built (apply launches skipped, stale data in the buffers): 17.09 -> 16.59 ms.
FUSE_BN_IN = os.environ.get('DUALVAR_FUSE_BN_IN', '1') != '0'         # (A/B switch; tests monkeypatch the module attribute)


class Slot:
    __slots__ = ('tensor', 'kind', 'off', 'size', 'Cout', 'Cin', 'taps', 'cin_pitch', 'cout_pitch', 'wd_off',
                 'shape', 'strides', 'parts', 'kw_store', 'w3_off', 'wd3_off')


class ParamStore:
    """Arenas for the trainable tensors of one encoder (registered in forward order)."""

    def __init__(self):
        self.slots = []
        self.merged = []           # virtual slots: several convs that share their input, stored back to back
        self._by_id = {}
        self.buffers = []          # (module, name) float buffers (BN running stats) -- left where they are
        self.nbt = []              # BN modules whose num_batches_tracked is bumped per forward
        self.master = self.grad = self.cc = self.wd = None
        self.dtype = None
        self.generation = 0
        self._dirty = True            # master changed: compute copy + dgrad layout stale
        self._cast_done = False       # ... but the bf16 copy was already refreshed by the update kernel
        self._versions = None
        self.total = 0
        self.no_dgrad = False         # key encoders (never back-propagated) skip the dgrad-layout weights
        self.pending_backward = 0     # forward passes with autograd history whose backward has not run yet (reset by the optimizer)
        self._fp8, self._fp8_ws, self._fp8_stale = {}, None, False    # fp8 copies of pointwise-conv weights (fp8_weights)

    # ---- registration (idempotent per tensor: S3D registers its stem twice)
    def _add(self, t, kind, **kw):
        if id(t) in self._by_id:
            return self._by_id[id(t)]
        s = Slot()
        s.tensor, s.kind = t, kind
        s.Cout = s.Cin = s.taps = s.cin_pitch = s.cout_pitch = 0
        s.wd_off = -1
        s.kw_store = 0
        for k, v in kw.items():
            setattr(s, k, v)
        self.slots.append(s)
        self._by_id[id(t)] = s
        return s

    def add_conv(self, weight, cin_pitch=None, need_dgrad=True, kw_store=0):
        """weight: [O, I, kt, kh, kw] or [O, I] parameter.  kw_store > kw stores each kernel row kw_store taps wide
        (the extra taps are structural zeros that the parameter view skips): the RGB stem, see Plan.conv."""
        O, I = weight.shape[:2]
        taps = 1
        for d in weight.shape[2:]:
            taps *= d
        if kw_store:
            assert weight.dim() == 5 and kw_store >= weight.shape[4] and not need_dgrad
            taps = taps // weight.shape[4] * kw_store
        return self._add(weight, 'conv', Cout=O, Cin=I, taps=taps, cin_pitch=cin_pitch or cp8(I), cout_pitch=cp8(O),
                         wd_off=0 if (need_dgrad and not self.no_dgrad) else -1, kw_store=kw_store)

    def add_merged(self, weights):
        """Several 1x1x1 convs reading the same input become ONE GEMM: their [Cout_i][Cin] blocks are registered
        back to back, so together they are a [sum Cout_i][Cin] matrix (forward / wgrad use it in place; the dgrad
        layout gets its own packed copy).  Returns the virtual slot."""
        parts = [self.add_conv(w, need_dgrad=False) for w in weights]
        key = tuple(id(w) for w in weights)
        for m in self.merged:
            if tuple(id(p.tensor) for p in m.parts) == key:
                return m
        assert all(p.taps == 1 and p.Cin == parts[0].Cin and p.Cout % 8 == 0 for p in parts)
        m = Slot()
        m.tensor, m.kind, m.parts = None, 'merged', parts
        m.Cin, m.taps, m.cin_pitch = parts[0].Cin, 1, parts[0].cin_pitch
        m.Cout = sum(p.Cout for p in parts)
        m.cout_pitch = cp8(m.Cout)
        m.wd_off = -1 if self.no_dgrad else 0
        m.off = m.size = 0
        self.merged.append(m)
        return m

    def add_vec(self, t):
        return self._add(t, 'vec')

    def add_bn(self, bn):
        self.add_vec(bn.weight)
        self.add_vec(bn.bias)
        if bn not in self.nbt:
            self.nbt.append(bn)

    def slot(self, t):
        return self._by_id[id(t)]

    # ---- materialisation
    def ready(self, device, dtype):
        if self.master is None or self.master.device != device or self.dtype != dtype:
            return False
        s0, s1 = self.slots[0], self.slots[-1]
        es = 4
        return (s0.tensor.data_ptr() == self.master.data_ptr() + s0.off * es and
                s1.tensor.data_ptr() == self.master.data_ptr() + s1.off * es)

    def _view(self, arena, s):
        if s.kind == 'vec':
            return arena.narrow(0, s.off, s.tensor.numel()).view(s.tensor.shape)
        return torch.as_strided(arena, s.shape, s.strides, s.off)

    def materialize(self, device, dtype):
        off = 0
        wd_off = 0
        for s in self.slots:
            s.off = off
            if s.kind == 'vec':
                s.size = _align8(s.tensor.numel())
            else:
                s.size = s.Cout * s.taps * s.cin_pitch
                shp = tuple(s.tensor.shape)
                rowp = s.taps * s.cin_pitch
                if len(shp) == 5:
                    kt, kh, kw = shp[2:]
                    kw = s.kw_store or kw
                    s.strides = (rowp, 1, kh * kw * s.cin_pitch, kw * s.cin_pitch, s.cin_pitch)
                else:
                    s.strides = (rowp, 1)
                s.shape = shp
                if s.wd_off >= 0:
                    s.wd_off = wd_off
                    wd_off += _align8(s.Cin * s.taps * s.cout_pitch)
            off += _align8(s.size)
        for m in self.merged:
            m.off, m.size = m.parts[0].off, sum(p.size for p in m.parts)
            for a, b in zip(m.parts, m.parts[1:]):
                assert b.off == a.off + a.size, 'merged convs must be registered back to back'
            if m.wd_off >= 0:
                m.wd_off = wd_off
                wd_off += _align8(m.Cin * m.taps * m.cout_pitch)
        self.total = off
        master = torch.zeros(off, dtype=torch.float32, device=device)
        grad = torch.zeros(off, dtype=torch.float32, device=device)
        with torch.no_grad():
            for s in self.slots:
                v = self._view(master, s)
                v.copy_(s.tensor.data.to(device=device, dtype=torch.float32))
                rg = s.tensor.requires_grad
                s.tensor.data = v
                s.tensor.grad = self._view(grad, s) if rg else None
        self.master, self.grad, self.dtype = master, grad, dtype
        self.cc = master if dtype == DV_F32 else torch.zeros(off, dtype=torch.bfloat16, device=device)
        self.wd = torch.zeros(max(wd_off, 8), dtype=ops.TORCH_DTYPE[dtype], device=device)
        self._w3_setup(device, dtype)
        # pack descriptors (device copies)
        packs = [s for s in self.slots + self.merged if s.kind in ('conv', 'merged') and s.wd_off >= 0]
        self._n_pack_blocks = 0
        if packs:
            arr = (L.PackDesc * len(packs))()
            bmap = []
            for i, s in enumerate(packs):
                a = arr[i]
                a.src_off, a.dst_off, a.Cout, a.Cin, a.taps, a.cin_pitch, a.cout_pitch = \
                    s.off, s.wd_off, s.Cout, s.Cin, s.taps, s.cin_pitch, s.cout_pitch
                bmap += [(i, c) for c in range(s.Cin)]
            self._pack_descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            self._pack_map = torch.tensor(bmap, dtype=torch.int32).to(device)
            self._n_pack_blocks = len(bmap)
        # BN counters in one int64 arena
        if self.nbt:
            self._nbt_arena = torch.zeros(len(self.nbt), dtype=torch.long, device=device)
            for i, bn in enumerate(self.nbt):
                if bn.num_batches_tracked is not None:
                    self._nbt_arena[i] = int(bn.num_batches_tracked)
                    bn.num_batches_tracked.data = self._nbt_arena[i]
                for nm in ('running_mean', 'running_var'):
                    b = getattr(bn, nm)
                    if b.device != device:
                        b.data = b.data.to(device)
        self.generation += 1
        self._dirty = True
        self._versions = None
        self._fp8, self._fp8_stale = {}, False

    def trainable_ranges(self):
        """[(first element, count)] of the maximal runs of arena slots whose tensors require gradients (alignment
        padding between trainable neighbours is included: it is zero and stays zero)"""
        sig = tuple(bool(s.tensor.requires_grad) for s in self.slots)
        if getattr(self, '_tr_sig', None) != (sig, self.generation):
            out, start, end = [], None, 0
            for s, rg in zip(self.slots, sig):
                if rg:
                    if start is None:
                        start = s.off
                    end = s.off + _align8(s.size)
                elif start is not None:
                    out.append((start, end - start))
                    start = None
            if start is not None:
                out.append((start, end - start))
            self._tr_sig, self._tr = (sig, self.generation), out
        return self._tr

    def attach_grads(self):
        """Re-attach .grad views (e.g. after zero_grad(set_to_none=True)); zeroes the arena if any was dropped.
        Called at the start of every backward: the full walk over the (hundreds of) slots only happens when the first
        and last trainable tensors show that views were dropped (torch drops all of them together) -- walking them
        every time left the GPU idle for ~70 us per call, twice per step."""
        trainable = [s for s in (self.slots[0], self.slots[-1]) if s.tensor.requires_grad] or \
            [s for s in self.slots if s.tensor.requires_grad][:1]
        if all(s.tensor.grad is not None for s in trainable):
            return
        dropped = False
        for s in self.slots:
            if s.tensor.requires_grad and s.tensor.grad is None:
                dropped = True
                s.tensor.grad = self._view(self.grad, s)
        if dropped:
            self.grad.zero_()

    def zero_grad(self):
        # bucket all-reduces started from inside a backward whose optimizer step never ran (parallel.GradSync._on_ready): wait
        # for them and re-arm, or every later backward would be refused as "gradient accumulation" 
        if getattr(self, '_sync_started', False):
            owner = getattr(self, '_sync_owner', None)
            if owner is not None:
                owner.abandon(self)
            self._sync_started = False
        # (pending_backward is NOT reset here: the reference's loop -- and ours -- calls zero_grad() between the forward and the
        # backward of a step, pretrain.py:447-448; the optimizer step resets it)
        if self.grad is not None:
            self.grad.zero_()

    def mark_dirty(self, cast_done=False):
        self._dirty = True
        self._cast_done = cast_done

    def _version_sum(self):
        return sum(s.tensor._version for s in self.slots)

    def refresh(self, check_versions=True):
        """Bring the compute-dtype copy and the dgrad-layout weights up to date with the master."""
        if not self._dirty and check_versions:
            v = self._version_sum()
            if v != self._versions:
                self._dirty = True
        if not self._dirty:
            if self._fp8_stale:
                self._refresh_fp8()
            return
        if self.dtype == DV_BF16 and not self._cast_done:
            ops.call('dv_cast_arena', DV_BF16, self.master, self.cc, self.total)
        if self._n_pack_blocks:
            ops.call('dv_pack_dgrad_weights', self.dtype, self.master, self.wd, self._pack_descs, self._pack_map,
                     self._n_pack_blocks)
        self._refresh_w3()
        self._refresh_fp8()
        self._dirty = False
        self._cast_done = False
        self._versions = self._version_sum()

    # ---- fp32 mode: weights pre-split into three bf16 in fragment order (include/dualvar_hip.h: dv_pack_w3, DV_W3)
    def _w3_setup(self, device, dtype):
        self.w3 = self.wd3 = None
        self._w3_jobs = []
        for s_ in self.slots + self.merged:
            s_.w3_off = s_.wd3_off = -1
        if dtype != DV_F32 or L.f32_exact():
            return
        lib = L.load()
        convs = [s_ for s_ in self.slots + self.merged if s_.kind in ('conv', 'merged')]

        def table(rows_k, src_of, attr):
            arr, bmap, off = (L.W3Desc * len(rows_k))(), [], 0
            for i, (s_, rows, ktot) in enumerate(rows_k):
                setattr(s_, attr, off)
                arr[i].src_off, arr[i].dst_off, arr[i].N, arr[i].Ktot = src_of(s_), off, rows, ktot
                nbytes = int(lib.dv_w3_bytes(rows, ktot))
                bmap += [(i, u) for u in range(0, nbytes // 48, 256)]
                off += nbytes
            return (torch.zeros(max(off, 16), dtype=torch.uint8, device=device),
                    torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device),
                    torch.tensor(bmap, dtype=torch.int32).to(device), len(bmap))
        fwd = [(s_, s_.Cout, s_.taps * s_.cin_pitch) for s_ in convs]
        self.w3, d1, m1, n1 = table(fwd, lambda s_: s_.off, 'w3_off')
        self._w3_jobs.append(('master', self.w3, d1, m1, n1))
        bwd = [(s_, s_.Cin, s_.taps * s_.cout_pitch) for s_ in convs if s_.wd_off >= 0]
        if bwd:
            self.wd3, d2, m2, n2 = table(bwd, lambda s_: s_.wd_off, 'wd3_off')
            self._w3_jobs.append(('wd', self.wd3, d2, m2, n2))

    def _refresh_w3(self):
        for src, out, descs, bmap, n in getattr(self, '_w3_jobs', ()):
            base = self.master if src == 'master' else self.wd
            ops.call('dv_pack_w3', base, out, descs, bmap, n)

    def fp8_weights(self, s):
        """(w8 [Cout][CinP] e4m3, scale, wd8 [Cin][CoutP] e4m3, scale) of a 1x1x1 conv slot for the fp8 pointwise path; the
        copies are re-quantised (per-tensor amax scaling) whenever the master weights change -- see refresh()"""
        e = self._fp8.get(id(s))
        if e is None:
            dev = self.master.device
            e = (s, torch.zeros(s.Cout, s.cin_pitch, dtype=torch.uint8, device=dev), torch.ones(1, device=dev),
                 torch.zeros(s.Cin, s.cout_pitch, dtype=torch.uint8, device=dev), torch.ones(1, device=dev))
            self._fp8[id(s)] = e
            self._fp8_stale = True
        return e[1:]

    def _refresh_fp8(self):
        if not self._fp8:
            return
        lib = L.load()
        if self._fp8_ws is None:
            self._fp8_ws = torch.empty(lib.dv_quantize_fp8_workspace() // 4, dtype=torch.float32, device=self.master.device)
        es = ops.ESIZE[self.dtype]
        for s, w8, sw, wd8, swd in self._fp8.values():
            ops.call('dv_quantize_fp8', self.dtype, self.cc.data_ptr() + s.off * es, s.Cout, s.cin_pitch, s.cin_pitch, 0, w8, s.cin_pitch,
                     sw, self._fp8_ws)
            if s.wd_off >= 0:
                ops.call('dv_quantize_fp8', self.dtype, self.wd.data_ptr() + s.wd_off * es, s.Cin, s.cout_pitch, s.cout_pitch, 0, wd8,
                         s.cout_pitch, swd, self._fp8_ws)
        self._fp8_stale = False

    # pointers
    def w_fwd(self, s):
        """-> (pointer, extra conv flags): the forward-layout weights of a slot for dv_conv3d_fwd"""
        if getattr(s, 'w3_off', -1) is not None and getattr(s, 'w3_off', -1) >= 0 and self.w3 is not None:
            return self.w3.data_ptr() + s.w3_off, L.DV_W3
        return self.cc.data_ptr() + s.off * ops.ESIZE[self.dtype], 0

    def w_master(self, s):
        return self.master.data_ptr() + s.off * 4

    def w_grad(self, s):
        return self.grad.data_ptr() + s.off * 4

    def w_dgrad(self, s, strided=False):
        """-> (pointer, extra conv flags) for dv_conv3d_dgrad (the pre-split form serves stride-1 problems only)"""
        if not strided and getattr(s, 'wd3_off', -1) is not None and getattr(s, 'wd3_off', -1) >= 0 and self.wd3 is not None:
            return self.wd3.data_ptr() + s.wd3_off, L.DV_W3
        return self.wd.data_ptr() + s.wd_off * ops.ESIZE[self.dtype], 0

    def bump_bn_counters(self):
        if self.nbt:
            self._nbt_arena += 1


class Comm:
    """Data-parallel context for SyncBatchNorm statistics (one process per GPU, RCCL)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # `exchange`: take the multi-rank code path.  DUALVAR_FORCE_EXCHANGE=1 takes it with ONE rank too: the only way
        # to run the RCCL calls (stream semantics, flat gather, async handles) on a single-GPU box, where two ranks
        # cannot share the device (tests/test_distributed_gpu.py::test_rccl_single_rank_rehearsal)
        self.exchange = self.world > 1 or (os.environ.get('DUALVAR_FORCE_EXCHANGE') == '1' and dist.is_available()
                                           and dist.is_initialized())
        # RCCL gathers straight into one flat tensor; gloo (the CPU / single-GPU tests) only has the list form
        self.flat_gather = self.exchange and dist.get_backend(group) == 'nccl'
        # RCCL enqueued directly on the step's own streams (dualvar_amd/rccl.py); None: torch.distributed calls
        self.rccl = rccl.get('bn', group) if self.flat_gather else None
        self._xstream = None

    def xstream(self, device):
        """the stream the backward sum exchanges run on, next to the main chain"""
        if self._xstream is None:
            self._xstream = torch.cuda.Stream(device=device)
        return self._xstream


class Launch:
    """One kernel launch of a plan: a bound C-ABI call plus its algorithmic cost (for the roofline report)."""
    __slots__ = ('name', 'kname', 'fn', 'args', 'bytes', 'flops', 'shape', 'gend')

    def __init__(self, name, kname, fn, args, nbytes=0, flops=0, shape=''):
        self.name, self.kname, self.fn, self.args, self.bytes, self.flops, self.shape = name, kname, fn, args, nbytes, flops, shape
        self.gend = 0                # end (element offset) of the highest gradient-arena range this launch writes

    def __call__(self, stream):
        rc = self.fn(*self.args, stream)
        if rc:
            L.check(rc, self.name)


class StreamStep:
    """A host-side step that needs the stream it is issued on (event hops around a direct RCCL call)."""
    __slots__ = ('name', 'kname', 'call', 'bytes', 'flops')

    def __init__(self, name, call):
        self.name, self.kname, self.call, self.bytes, self.flops = name, 'host:' + name, call, 0, 0

    def __call__(self, stream):
        self.call(stream)


class HostStep:
    """A host-side step inside a plan (a torch.distributed collective on the current stream)."""
    __slots__ = ('name', 'kname', 'call', 'bytes', 'flops')

    def __init__(self, name, call):
        self.name, self.kname, self.call, self.bytes, self.flops = name, 'host:' + name, call, 0, 0

    def __call__(self, stream):
        self.call()


def _conv_kname(lib, d, dgrad, dt, gv=16):
    """label of the kernel dv_conv3d_fwd / dv_conv3d_dgrad will launch for this descriptor (bench.py / profiles group by it)"""
    mode = 'DGRAD' if dgrad else 'FWD'
    tap = int(lib.dv_conv3d_tap_kind(C.byref(d), int(dgrad)))
    if tap == 3:
        return 'conv_pp<%s,%s>' % (dt, mode)         # (the pixel-pair stem form: tiles of whole output lines)
    if tap:
        return 'conv_tap<%s,%s,%s,%d,64>' % (dt, mode, 'sp' if tap == 1 else 'tm', int(lib.dv_conv3d_tap_rows(C.byref(d), int(dgrad))))
    ks = int(lib.dv_conv3d_ksplit_cols(C.byref(d), int(dgrad)))
    if ks:
        return 'conv_gemm_ks<%s,%s,64,%d>' % (dt, mode, ks)
    return 'conv_gemm<%s,%s,%d,%d,%d>' % ((dt, mode, gv) + _tile_shape(lib, d, dgrad))


def _tile_shape(lib, d, dgrad):
    """(rows, cols) of the GEMM tile the library will pick: only for the per-launch profiling labels."""
    r, c = C.c_int32(0), C.c_int32(0)
    L.check(lib.dv_conv3d_tile_shape(C.byref(d), int(dgrad), C.byref(r), C.byref(c)), 'dv_conv3d_tile_shape')
    return r.value, c.value


def _dt(dtype):
    return 'f32' if dtype == DV_F32 else 'bf16'


class Plan:
    """Static forward/backward launch lists for one backbone at one input shape."""

    def __init__(self, store, dtype, device, with_grad=True, comm=None, training=True):
        self.store, self.dtype, self.device, self.with_grad = store, dtype, device, with_grad
        self.training = training     # False: eval-mode BatchNorm (running statistics), forward only
        assert training or not with_grad, 'eval-mode plans are forward only' 
        self.comm = comm if comm is not None else Comm()
        self.ops = []
        self.f_list, self.b_list = [], []
        self.bytes = 0
        self.lib = L.load()
        self.timer = None            # set by bench.py: callable(launch, stream) recording HIP events
        self._zero_words = 0         # fp32 words that must be zero at the start of every backward (atomic targets)
        self.zero_arena = None
        self._side = None            # side stream + events of run_backward
        self.wgrad_early = False     # large weight gradients leave for the side stream at once (finalize decides)
        self._events = None
        self.grad_ready = None       # callable(plan, lo): gradient-arena elements [lo, total) are final (GradSync.attach)
        self.fp8_pointwise = False   # compute mode 'fp8pw' (backbone/base.py: set_compute_dtype)
        self._fp8_ws = None
        self.bucket_starts = ()      # ... called when lo drops to / below each of these element offsets
        self._triggers = None

    # ------------------------------------------------------------------ buffers
    def act(self, N, T, H, W, C_, dtype=None, cpitch=None, grad=None, zero=False):
        dtype = self.dtype if dtype is None else dtype
        a = ops.new_act(N, T, H, W, C_, dtype, self.device, cpitch=cpitch, zero=zero)
        self.bytes += a.buf.numel() * a.buf.element_size()
        want_grad = self.with_grad if grad is None else grad
        if want_grad:
            a.grad = ops.new_act(N, T, H, W, C_, dtype, self.device, cpitch=cpitch, zero=True)
            self.bytes += a.buf.numel() * a.buf.element_size()
        return a

    def slice(self, a, off, C_):
        s = a.slice(off, C_)
        if a.grad is not None:
            s.grad = a.grad.slice(off, C_)
        return s

    def f32(self, *shape, zero=True):
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=self.device)
        self.bytes += t.numel() * 4
        return t

    def reserve_zero(self, n):
        off = self._zero_words
        self._zero_words += (n + 7) & ~7
        return off

    def zero_ptr(self, off):
        return self.zero_arena.data_ptr() + off * 4

    # ------------------------------------------------------------------ graph ops
    def _push(self, op):
        self.ops.append(op)
        return op

    def conv(self, slot, x, k, s, p, out=None, stats=True, fp8=False):
        fp8 = bool(fp8 and getattr(self, 'fp8_pointwise', False) and self.dtype == DV_BF16 and tuple(k) == (1, 1, 1)
                   and tuple(s) == (1, 1, 1) and tuple(p) == (0, 0, 0) and slot.cin_pitch % 16 == 0 and slot.cout_pitch % 16 == 0
                   and slot.kind == 'conv' and x.cpitch == slot.cin_pitch)
        if getattr(x, 'hw_pad', 0):
            # RGB stem on the zero-bordered ingest frames (include/dualvar_hip.h, dv_ingest_ncdhw_pad): the 7x7
            # stride-2 padding-3 conv becomes a (kt,7,4)-tap stride-(st,2,1) conv over 8-channel pixel pairs
            assert k[1:] == (7, 7) and s[1:] == (2, 2) and p[1:] == (3, 3) and x.hw_pad == 3 and slot.kw_store == 8, \
                'the bordered ingest layout is for 7x7 / stride 2 / padding 3 RGB stems'
            pairs = Act(x.buf.view(-1, 8), x.N, x.T, x.H, x.W // 2, 8, 8, 0, x.dtype, 8)
            view = Slot()
            for f in Slot.__slots__:
                setattr(view, f, getattr(slot, f, None))
            view.cin_pitch, view.Cin = 8, 8
            op = self._push(ConvOp(self, view, pairs, (k[0], 7, 4), (s[0], 2, 1), (p[0], 0, 0), out, stats))
            op.alg_k = k[0] * 49 * 3                     # algorithmic K for the flop count (not the padded 224)
            op.zero_pad_taps = (slot.Cout * k[0] * 7, 32, 28, 4)    # rows, pitch, first pad column, count
            op.y.producer = op
            return op.y
        op = self._push(ConvOp(self, slot, x, k, s, p, out, stats, fp8=fp8))
        op.y.producer = op
        return op.y

    def bn_group(self, specs):
        """specs: [(bn_module, raw_conv_output, relu, residual_or_None, out_or_None), ...] of mutually independent
        layers -> their outputs.  One statistics exchange for the whole group."""
        op = self._push(BNGroupOp(self, specs))
        for m in op.members:
            m.y.bn_member = (op, m)
        return [m.y for m in op.members]

    def bn(self, bn_mod, x, relu=True, residual=None, out=None, conv_bias=None):
        """conv_bias: the bias vector of the conv that produced x (c3d.py: Conv3d(bias=True) + BatchNorm3d).  The conv
        kernel runs bias-free: in train mode the bias cancels in the normalised output and only shifts the running mean,
        in eval mode it folds into the shift; its gradient is identically zero (weight decay still applies)."""
        ys = self.bn_group([(bn_mod, x, relu, residual, out)])
        self.ops[-1].members[0].conv_bias = conv_bias
        return ys[0]

    def maxpool(self, x, k, s, p, sole_consumer=False):
        """sole_consumer: the caller guarantees that nothing else reads x.  If x is then the output of a lone
        BatchNorm + ReLU, the pair runs fused (dv_bn_apply_maxpool / dv_bn_bwd_*_maxpool): x is never materialised."""
        member = None
        if sole_consumer and FUSE_BN_POOL and getattr(x, 'bn_member', None) is not None:
            op, m = x.bn_member
            if (len(op.members) == 1 and m.relu and m.res is None and m.conv_bias is None and m.fused_pool is None
                    and m.y is x and x.off == 0 and (not self.with_grad or m.mask_from_x)):
                member = m
        return self._push(PoolOp(self, x, k, s, p, bn_member=member)).y

    def gate_group(self, fcs, cat):
        """in-place self gating of a concat buffer: fcs = [(nn.Linear, channel offset, width), ...]"""
        self._push(GateGroupOp(self, fcs, cat))
        return cat

    def spatial_mean(self, x):
        return self._push(MeanOp(self, x)).out

    def finalize(self):
        for op in self.ops:          # a BatchNorm output fused into its pool does not exist in memory: nobody else may read it
            if isinstance(op, PoolOp) and op.bn_member is not None:
                gone = op.bn_member.y
                for other in self.ops:
                    if other is op:
                        continue
                    readers = [getattr(other, a, None) for a in ('x', 'cat')]
                    readers += [t for m in getattr(other, 'members', ()) for t in (m.x, m.res)]
                    assert all(r is None or r.buf is not gone.buf for r in readers), \
                        'maxpool(sole_consumer=True) on an activation that %s also reads' % type(other).__name__
        # the last consumer (forward order) of an activation is the first writer of its gradient
        seen = set()
        for op in reversed(self.ops):
            for name, a in op.grad_targets():
                g = a.grad if a.grad is not None else a      # (an activation fused away has no buffer of its own: key on the gradient)
                key = (g.buf.data_ptr(), g.off)
                op.acc[name] = key in seen
                seen.add(key)
        # conv -> BatchNorm(+ReLU) -> conv chains: when the second conv is the only reader of the BatchNorm's output, its
        # data gradient IS dL/dy of that BatchNorm, complete after one launch -- the BatchNorm backward's reduce (sum g,
        # sum g*xhat) runs in that launch's epilogue (dv_conv3d_dgrad_bn) instead of re-reading dL/dy in a pass of its own
        if self.with_grad and self.training and FUSE_BN_REDUCE:
            writers = {}
            for op in self.ops:
                for name, a in op.grad_targets():
                    g = a.grad if a.grad is not None else a
                    writers.setdefault((g.buf.data_ptr(), g.off), []).append((op, a))
            for op in self.ops:
                if not isinstance(op, BNGroupOp):
                    continue
                for m in op.members:
                    y = m.y
                    if y.grad is None or m.res is not None or m.fused_pool is not None or (m.relu and not m.mask_from_x):
                        continue
                    w = writers.get((y.grad.buf.data_ptr(), y.grad.off), [])
                    if len(w) != 1 or not isinstance(w[0][0], ConvOp):
                        continue
                    cop, a = w[0]
                    if (cop.fp8 or not cop.need_dx or cop.acc.get('x') or a.buf is not y.buf or a.off != y.off or a.C != y.C
                            or a.rows != y.rows or cop.bn_fuse is not None):
                        continue
                    cop.bn_fuse, m.reduce_fused = m, True
        if self.training and FUSE_BN_IN and self.dtype == DV_F32:
            for op in self.ops:
                if not isinstance(op, BNGroupOp):
                    continue
                for m in op.members:
                    if (m.res is not None or m.fused_pool is not None or m.conv_bias is not None
                            or (self.with_grad and m.relu and not m.mask_from_x)):
                        continue
                    y = m.y
                    readers = [o for o in self.ops if o is not op and any(
                        r is not None and r.buf is y.buf for r in [getattr(o, 'x', None), getattr(o, 'cat', None)] +
                        [t for mm in getattr(o, 'members', ()) for t in (mm.x, mm.res)])]
                    if len(readers) != 1 or not isinstance(readers[0], ConvOp):
                        continue
                    cop = readers[0]
                    if (cop.fp8 or cop.bn_in is not None or cop.x.buf is not y.buf or cop.x.off != y.off or cop.x.C != y.C
                            or m.x.cpitch != cop.slot.cin_pitch or m.x.rows != y.rows or m.x.C != y.C):
                        continue
                    d = ops.conv_desc(cop.dtype, m.x, cop.y, cop.k, cop.s, cop.p, flags=cop.d.flags)
                    if not int(self.lib.dv_conv3d_bn_in_ok(C.byref(d))) or ops.tile_rows(d) != cop.tile_rows:
                        continue
                    cop.bn_in, m.fused_conv = m, cop
        self.bn_fuse_ws = None
        if self.with_grad and self.training and FUSE_BN_REDUCE_TAP and self.dtype == DV_F32 and not FUSE_BN_REDUCE:
            writers, need_ws = {}, 0
            for op in self.ops:
                for name, a in op.grad_targets():
                    g = a.grad if a.grad is not None else a
                    writers.setdefault((g.buf.data_ptr(), g.off), []).append((op, a))
            for op in self.ops:
                if not isinstance(op, BNGroupOp):
                    continue
                for m in op.members:
                    y = m.y
                    if y.grad is None or m.res is not None or m.fused_pool is not None or (m.relu and not m.mask_from_x):
                        continue
                    w = writers.get((y.grad.buf.data_ptr(), y.grad.off), [])
                    if len(w) != 1 or not isinstance(w[0][0], ConvOp):
                        continue
                    cop, a = w[0]
                    if (cop.fp8 or not cop.need_dx or cop.acc.get('x') or a.buf is not y.buf or a.off != y.off or a.C != y.C
                            or a.rows != y.rows or cop.bn_fuse is not None or m.x.ld * 4 * (m.x.rows - 1) >= (1 << 31)):
                        continue
                    _, wdflag = self.store.w_dgrad(cop.slot, strided=False)
                    if not wdflag or max(cop.s) > 1 or cop.slot.cout_pitch * cop.k[0] * cop.k[1] * cop.k[2] < FUSE_BN_REDUCE_TAP_MIN_K:
                        continue                 # (short K loops: the epilogue's read of the BatchNorm input is not hidden)
                    d3 = ops.conv_desc(cop.dtype, cop.x, cop.y, cop.k, cop.s, cop.p, flags=wdflag)
                    nb = int(self.lib.dv_conv3d_dgrad_bn_workspace(C.byref(d3)))
                    if nb <= 0:
                        continue
                    cop.bn_fuse, cop.bn_fuse_tap, m.reduce_fused = m, True, True
                    need_ws = max(need_ws, nb)
            if need_ws:
                # one workspace for all of them (they run one after the other on the main stream; the ticket words are zero
                # between launches, and after a failed launch: _lib.register_ticket_workspace)
                self.bn_fuse_ws = L.register_ticket_workspace(self.f32(need_ws // 4))
        if self.with_grad and self.training and FUSE_BN_WGRAD and self.dtype == DV_F32:
            for op in self.ops:
                if not isinstance(op, BNGroupOp) or len(op.members) != 1:
                    continue
                m = op.members[0]
                cop = m.conv
                if (not isinstance(cop, ConvOp) or cop.need_dx or cop.fp8 or cop.bn_apply is not None or m.res is not None
                        or (m.relu and not m.mask_from_x) or m.x.buf is not cop.y.buf or m.x.off != cop.y.off
                        or m.x.C != cop.y.C or m.x.grad is None or m.y.grad is None or m.x.ld != m.y.grad.ld):
                    continue
                d = ops.conv_desc(cop.dtype, cop.x, cop.y, cop.k, cop.s, cop.p, flags=0)
                if d.ldy != m.y.grad.ld or not self.lib.dv_conv3d_wgrad_bn_ok(C.byref(d)):
                    continue
                cop.bn_apply, m.apply_fused = m, True
        # scratch of the deterministic weight gradients (row-split partial tiles): ONE buffer per plan, sized for the
        # largest layer -- the plan's weight gradients all run on one stream (the side stream), each followed by its
        # reduce, so they can share it
        self.wgrad_ws = None
        if self.with_grad:
            need = max([o.wgrad_workspace_bytes() for o in self.ops if isinstance(o, ConvOp)] + [0])
            if need:
                self.wgrad_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                self.bytes += need
        if self.with_grad and self._zero_words:
            self.zero_arena = self.f32(self._zero_words)
            arena = self.zero_arena
            self.b_list.append(HostStep('zero_bwd_sums', arena.zero_))
        for op in self.ops:
            f, b = op.launches()
            self.f_list += f
            op._b = b
        for op in reversed(self.ops):
            self.b_list += op._b
        self.b_list = overlap_bn_exchange(self.b_list)
        # early release of the large weight gradients: see WGRAD_EARLY_RATIO
        t_conv = sum(max(l.flops / 1.5e14, l.bytes / 4e12) for l in self.b_list if l.name == 'conv_dgrad')
        t_other = sum(l.bytes / 4e12 for l in self.b_list if l.name != 'conv_dgrad' and l.name not in SIDE_LAUNCHES)
        self.wgrad_early = (t_other >= WGRAD_EARLY_RATIO * t_conv) if WGRAD_EARLY == 'auto' else WGRAD_EARLY == '1'

    # ------------------------------------------------------------------ execution
    def _run(self, lst):
        s = ops.stream_ptr()
        t = self.timer
        if t is None:
            for l in lst:
                l(s)
        else:
            for l in lst:
                t(l, s)

    def run_forward(self):
        # (Independent convs of an Inception level on a second stream, forked / joined with one event each way -- 27 pairs per pass --
        # were measured again in round 4, with the LDS-staged kernels that can share a CU: 16.97 -> 17.10 ms, three A/B pairs.  The
        # event hops cost more than the overlap of two 20 - 70 us kernels gains.  Not kept.)
        self._run(self.f_list)

    def run_backward(self):
        """Weight gradients leave the critical path: a conv's wgrad needs only the finished gradient of its own
        output and a forward activation, and nothing reads its result before the optimizer -- so every conv_wgrad
        (and the stem's pad-tap clear that follows it) is issued on a SIDE stream behind an event recorded on the main
        stream at its list position, and the main stream joins the side stream at the end.  The dgrad / BatchNorm /
        pool chain of the small late layers leaves most CUs idle; the wgrads fill them."""
        if not WGRAD_SIDE_STREAM or not self.b_list:
            return self._run(self.b_list)
        trig = self._grad_triggers() if self.grad_ready is not None else {}
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._events = [torch.cuda.Event() for l in self.b_list if l.name in SIDE_LAUNCHES]
        side = self._side
        sm, ss = main.cuda_stream, side.cuda_stream
        t = self.timer
        k = 0
        pending = []                 # side-stream launches waiting for the next event (one event serves WGRAD_BATCH of them)
        last_side = False

        def flush():
            nonlocal k
            if not pending:
                return
            ev = self._events[k % len(self._events)]
            k += 1
            ev.record(main)
            side.wait_event(ev)
            for q in pending:
                if t is None:
                    q(ss)
                else:
                    t(q, ss, side)
            pending.clear()
        early = self.wgrad_early
        for l in self.b_list:
            nm = l.name
            if nm in SIDE_LAUNCHES or (nm == 'stem_pad_taps' and last_side):
                pending.append(l)
                last_side = True
                if len(pending) >= WGRAD_BATCH or (early and l.flops >= WGRAD_FLUSH_FLOPS):
                    flush()
            else:
                last_side = False
                if t is None:
                    l(sm)
                else:
                    t(l, sm)
            if trig:
                lo = trig.get(id(l))
                if lo is not None:
                    flush()
                    self.grad_ready(self, lo)
        flush()
        main.wait_stream(side)

    def _grad_triggers(self):
        """{id(launch): lo}: once that launch has been issued, no later launch of the backward list writes the
        gradient arena at or above element lo -- for the bucket starts GradSync asked about (highest first)."""
        key = tuple(self.bucket_starts)
        if self._triggers is None or self._triggers[0] != key:
            trig, remaining = {}, sorted(key, reverse=True)
            frontier = 0                                   # max gend over launches AFTER position i, built backwards
            after = [0] * len(self.b_list)
            for i in range(len(self.b_list) - 1, -1, -1):
                after[i] = frontier
                frontier = max(frontier, getattr(self.b_list[i], 'gend', 0))
            for i, l in enumerate(self.b_list):
                while remaining and after[i] <= remaining[0]:
                    trig[id(l)] = remaining.pop(0)          # several buckets may become final at once: keep the lowest
            self._triggers = (key, trig)
        return self._triggers[1]

    def cost(self):
        """(algorithmic bytes, flops) of one forward+backward replay."""
        ls = self.f_list + self.b_list
        return sum(l.bytes for l in ls), sum(l.flops for l in ls)


def _gend(*slots):
    """highest gradient-arena element a launch writing these slots touches (+1)"""
    return max(s.off + (s.size if s.kind != 'vec' else s.tensor.numel()) for s in slots)


def overlap_bn_exchange(b_list):
    """Reorder a backward launch list so that every SyncBN sum exchange overlaps the weight-gradient kernels issued
    since the previous exchange: [.. wgrad_a .. wgrad_b .. reduce, START, WAIT, apply ..] becomes
    [.. reduce, START, wgrad_a, wgrad_b, WAIT, apply ..].  Safe because a weight gradient only reads a forward
    activation and the (already final) gradient of its own conv output, and nothing in a backward list writes either
    again; its own output (the fp32 gradient arena) is not read before the optimizer."""
    out, since = [], 0                       # `since`: index in `out` just after the previous WAIT
    for l in b_list:
        if l.name == 'syncbn_allreduce_start':
            moved = [x for x in out[since:] if x.name == 'conv_wgrad']
            if moved:
                out[since:] = [x for x in out[since:] if x.name != 'conv_wgrad']
            out.append(l)
            out += moved
        else:
            out.append(l)
            if l.name == 'syncbn_allreduce_wait':
                since = len(out)
    return out


class Op:
    def __init__(self, plan):
        self.plan = plan
        self.acc = {}

    def grad_targets(self):
        return []

    def launches(self):
        """-> (forward launches, backward launches) in execution order"""
        raise NotImplementedError


def _abytes(a, c=None):
    return a.rows * (a.C if c is None else c) * ops.ESIZE[a.dtype]


class ConvOp(Op):
    """conv (bias-free, BN partial statistics in the epilogue) with wgrad + dgrad."""

    def __init__(self, plan, slot, x, k, s, p, out, stats, fp8=False):
        super().__init__(plan)
        self.slot, self.x, self.k, self.s, self.p = slot, x, k, s, p
        self.fp8 = fp8
        self.dtype = plan.dtype
        To, Ho, Wo = ops.conv_out_dims(x, k, s, p)
        self.y = out if out is not None else plan.act(x.N, To, Ho, Wo, slot.Cout)
        assert x.cpitch == slot.cin_pitch, (x.cpitch, slot.cin_pitch)
        stats = stats and plan.training              # eval-mode BatchNorm needs no batch statistics
        self._wf, wflag = plan.store.w_fwd(slot)
        self.d = ops.conv_desc(self.dtype, x, self.y, k, s, p, flags=(DV_STATS if stats else 0) | wflag)
        self.tiles = ops.stat_tiles(self.d)
        self.tile_rows = ops.tile_rows(self.d)
        self.stats = plan.f32(2, slot.Cout, self.tiles) if stats else None
        self.need_dx = plan.with_grad and x.grad is not None and slot.wd_off >= 0
        self.alg_k = None
        self.zero_pad_taps = None
        self.bn_fuse = None          # BNMember whose backward reduce this conv's data gradient carries (Plan.finalize)
        self.bn_fuse_tap = False     # ... in the ordered form of the LDS-staged kernel (dv_conv3d_dgrad_bn_ws)
        self.bn_apply = None         # BNMember (of this conv's output) whose backward apply this conv's weight gradient carries
        self.bn_in = None            # BNMember (of this conv's INPUT) applied on load: x is never materialised (FUSE_BN_IN)

    def grad_targets(self):
        return [('x', self.x)] if self.need_dx else []

    def wgrad_workspace_bytes(self):
        d = ops.conv_desc(self.dtype, self.x, self.y, self.k, self.s, self.p, flags=0)
        return ops.wgrad_workspace_bytes(d)

    def launches(self):
        p, st, lib, sl, x, y = self.plan, self.plan.store, self.plan.lib, self.slot, self.x, self.y
        es = ops.ESIZE[self.dtype]
        taps = self.k[0] * self.k[1] * self.k[2]
        kdim = self.alg_k if self.alg_k is not None else taps * sl.Cin
        flops = 2 * y.rows * sl.Cout * kdim
        wbytes = sl.Cout * kdim * es
        gv = 8 if (self.dtype == DV_BF16 and sl.cin_pitch % 8) else 16
        kf = _conv_kname(lib, self.d, 0, _dt(self.dtype), gv)
        shp = 'M%d Cin%d Cout%d k%s s%s' % (y.rows, sl.Cin, sl.Cout, 'x'.join(map(str, self.k)), 'x'.join(map(str, self.s)))
        f = [Launch('conv_fwd', kf, lib.dv_conv3d_fwd,
                    (C.byref(self.d), x.ptr, self._wf, 0, y.ptr, self.stats.data_ptr() if self.stats is not None else 0),
                    _abytes(x) + wbytes + _abytes(y), flops, shp)]
        if self.bn_in is not None:
            m = self.bn_in
            r = self._bn_in = L.BnIn()
            r.scale, r.shift, r.flags = m.scale.data_ptr(), m.shift.data_ptr(), (DV_RELU if m.relu else 0)
            self.d_in = ops.conv_desc(self.dtype, m.x, y, self.k, self.s, self.p, flags=self.d.flags)
            f = [Launch('conv_fwd', _conv_kname(lib, self.d_in, 0, _dt(self.dtype), gv) + '+bn_in', lib.dv_conv3d_fwd_bn_in,
                        (C.byref(self.d_in), m.x.ptr, C.byref(r), self._wf, y.ptr, self.stats.data_ptr() if self.stats is not None else 0),
                        _abytes(x) + wbytes + _abytes(y), flops, shp + ' +bn_in')]
        if self.fp8:
            f = self._fp8_forward(shp, flops, wbytes)
        b = []
        if p.with_grad:
            self.d_w = ops.conv_desc(self.dtype, x, y, self.k, self.s, self.p, flags=0)
            tr, tc, tsp = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            L.check(lib.dv_conv3d_wgrad_tile(C.byref(self.d_w), C.byref(tr), C.byref(tc), C.byref(tsp)), 'dv_conv3d_wgrad_tile')
            tile = '%d,%d' % (tr.value, tc.value)
            ws = p.wgrad_ws
            if self.bn_apply is not None:
                m = self.bn_apply
                gs, bs = st.slot(m.bn.weight), st.slot(m.bn.bias)
                r = self._bn_bwd = L.BnBwd()
                r.x, r.ldx = y.ptr, y.ld
                r.mean, r.invstd, r.scale, r.shift = (t.data_ptr() for t in (m.mean, m.invstd, m.scale, m.shift))
                r.gamma, r.dgamma, r.dbeta = st.w_master(gs), st.w_grad(gs), st.w_grad(bs)
                r.sums, r.n_rep, r.flags = p.zero_ptr(m.sums_off), BN_REPLICAS, (0 if m.relu else DV_NO_RELU_MASK)
                r.inv_count, r.dparam_scale = 1.0 / (m.M * p.comm.world), 1.0 / p.comm.world
                b.append(Launch('conv_wgrad', 'conv_wgrad<%s,%d,%s>+bn_bwd_apply' % (_dt(self.dtype), gv, tile), lib.dv_conv3d_wgrad_bn,
                                (C.byref(self.d_w), x.ptr, m.y.grad.ptr, st.w_grad(sl), ws.data_ptr() if ws is not None else 0,
                                 ws.numel() if ws is not None else 0, C.byref(r)),
                                _abytes(x) + 2 * _abytes(y) + sl.Cout * kdim * 4, flops, shp + ' +bn_bwd_apply'))
                b[-1].gend = max(sl.off + sl.size, _gend(gs, bs))
            elif self.bn_in is not None:
                m = self.bn_in
                self.d_w_in = ops.conv_desc(self.dtype, m.x, y, self.k, self.s, self.p, flags=0)
                b.append(Launch('conv_wgrad', 'conv_wgrad<%s,%d,%s>+bn_in' % (_dt(self.dtype), gv, tile), lib.dv_conv3d_wgrad_bn_in,
                                (C.byref(self.d_w_in), m.x.ptr, C.byref(self._bn_in), y.grad.ptr, st.w_grad(sl),
                                 ws.data_ptr() if ws is not None else 0, ws.numel() if ws is not None else 0),
                                _abytes(x) + _abytes(y) + sl.Cout * kdim * 4, flops, shp + ' +bn_in'))
                b[-1].gend = sl.off + sl.size
            else:
                b.append(Launch('conv_wgrad', 'conv_wgrad<%s,%d,%s>' % (_dt(self.dtype), gv, tile), lib.dv_conv3d_wgrad,
                                (C.byref(self.d_w), x.ptr, y.grad.ptr, st.w_grad(sl), ws.data_ptr() if ws is not None else 0,
                                 ws.numel() if ws is not None else 0),
                                _abytes(x) + _abytes(y) + sl.Cout * kdim * 4, flops, shp))
                b[-1].gend = sl.off + sl.size
            if self.zero_pad_taps is not None:
                rows, pitch, c0, nc = self.zero_pad_taps
                b.append(Launch('stem_pad_taps', 'fill_cols', lib.dv_fill_cols_f32, (st.w_grad(sl), rows, pitch, c0, nc, 0.0)))
                b[-1].gend = sl.off + sl.size
            if self.need_dx and self.fp8:
                b += self._fp8_dgrad(shp, flops, wbytes)
            elif self.need_dx:
                acc = bool(self.acc.get('x'))
                wdp, wdflag = st.w_dgrad(sl, strided=max(self.s) > 1)
                if max(self.s) > 1 and (self.bn_fuse is None or self.bn_fuse_tap):
                    # a strided data gradient takes the pre-split weights only where every parity class of it runs on the
                    # LDS-staged input-tile kernel (the 7x1x1 / stride-2 stem conv: dv_conv3d_tap_kind says so)
                    wdp3, wdflag3 = st.w_dgrad(sl, strided=False)
                    if wdflag3:
                        d3 = ops.conv_desc(self.dtype, x, y, self.k, self.s, self.p, flags=(DV_ACCUM if acc else 0) | wdflag3)
                        if int(lib.dv_conv3d_tap_kind(C.byref(d3), 1)):
                            wdp, wdflag = wdp3, wdflag3
                self.d_g = ops.conv_desc(self.dtype, x, y, self.k, self.s, self.p, flags=(DV_ACCUM if acc else 0) | wdflag)
                kd = _conv_kname(lib, self.d_g, 1, _dt(self.dtype))
                if self.bn_fuse is not None:
                    m = self.bn_fuse
                    r = self._bn_reduce = L.BnReduce()
                    r.x, r.ldx = m.x.ptr, m.x.ld
                    r.mean, r.invstd, r.scale, r.shift = (t.data_ptr() for t in (m.mean, m.invstd, m.scale, m.shift))
                    r.sums, r.n_rep, r.flags = p.zero_ptr(m.sums_off), BN_REPLICAS, (0 if m.relu else DV_NO_RELU_MASK)
                    if self.bn_fuse_tap:
                        ws = p.bn_fuse_ws
                        b.append(Launch('conv_dgrad', kd + '+bn_reduce', lib.dv_conv3d_dgrad_bn_ws,
                                        (C.byref(self.d_g), y.grad.ptr, wdp, x.grad.ptr, C.byref(r), ws.data_ptr(), ws.numel() * 4),
                                        _abytes(y) + wbytes + _abytes(x) * 2, flops, shp + ' +bn_reduce'))
                    else:
                        b.append(Launch('conv_dgrad', kd, lib.dv_conv3d_dgrad_bn, (C.byref(self.d_g), y.grad.ptr, wdp, x.grad.ptr, C.byref(r)),
                                        _abytes(y) + wbytes + _abytes(x) * 2, flops, shp + ' +bn_reduce'))
                else:
                    b.append(Launch('conv_dgrad', kd, lib.dv_conv3d_dgrad, (C.byref(self.d_g), y.grad.ptr, wdp, x.grad.ptr),
                                    _abytes(y) + wbytes + _abytes(x) * (2 if acc else 1), flops, shp))
        return f, b


def _fp8_methods():
    """fp8 pointwise path of ConvOp (compute mode 'fp8pw'; include/dualvar_hip.h: dv_conv3d_fwd_fp8 / dv_conv3d_dgrad_fp8)"""
    def _ws(self):
        p = self.plan
        if p._fp8_ws is None:
            p._fp8_ws = p.f32(p.lib.dv_quantize_fp8_workspace() // 4)
        return p._fp8_ws

    def _fp8_forward(self, shp, flops, wbytes):
        p, st, lib, sl, x, y = self.plan, self.plan.store, self.plan.lib, self.slot, self.x, self.y
        w8, sw, _, _ = st.fp8_weights(sl)
        self.x8 = torch.empty(x.rows, x.cpitch, dtype=torch.uint8, device=p.device)
        self.sx = p.f32(1)
        p.bytes += self.x8.numel()
        ws = _ws(self)
        self.d8 = ops.conv_desc(DV_BF16, x, y, self.k, self.s, self.p, flags=self.d.flags)
        self.d8.ldx = x.cpitch
        tile = _tile_shape(lib, self.d, 0)
        return [Launch('quantize_fp8', 'quantize_fp8<e4m3>', lib.dv_quantize_fp8,
                       (DV_BF16, x.ptr, x.rows, x.cpitch, x.ld, 0, self.x8.data_ptr(), x.cpitch, self.sx.data_ptr(), ws.data_ptr()),
                       2 * _abytes(x) + x.rows * x.cpitch, 0, shp),
                Launch('conv_fwd', 'conv_gemm<fp8,FWD,16,%d,%d>' % tile, lib.dv_conv3d_fwd_fp8,
                       (C.byref(self.d8), self.x8.data_ptr(), w8.data_ptr(), self.sx.data_ptr(), sw.data_ptr(), y.ptr,
                        self.stats.data_ptr() if self.stats is not None else 0),
                       x.rows * x.cpitch + wbytes // 2 + _abytes(y), flops, shp)]

    def _fp8_dgrad(self, shp, flops, wbytes):
        p, st, lib, sl, x, y = self.plan, self.plan.store, self.plan.lib, self.slot, self.x, self.y
        _, _, wd8, swd = st.fp8_weights(sl)
        self.dy8 = torch.empty(y.rows, y.cpitch, dtype=torch.uint8, device=p.device)
        self.sdy = p.f32(1)
        p.bytes += self.dy8.numel()
        ws = _ws(self)
        acc = bool(self.acc.get('x'))
        self.d8g = ops.conv_desc(DV_BF16, x, y, self.k, self.s, self.p, flags=DV_ACCUM if acc else 0)
        self.d8g.ldy = y.cpitch
        g = y.grad
        return [Launch('quantize_fp8', 'quantize_fp8<e5m2>', lib.dv_quantize_fp8,
                       (DV_BF16, g.ptr, g.rows, g.cpitch, g.ld, 1, self.dy8.data_ptr(), y.cpitch, self.sdy.data_ptr(), ws.data_ptr()),
                       2 * _abytes(y) + y.rows * y.cpitch, 0, shp),
                Launch('conv_dgrad', 'conv_gemm<fp8,DGRAD,16,%d,%d>' % _tile_shape(lib, self.d8g, 1), lib.dv_conv3d_dgrad_fp8,
                       (C.byref(self.d8g), self.dy8.data_ptr(), wd8.data_ptr(), self.sdy.data_ptr(), swd.data_ptr(), x.grad.ptr),
                       y.rows * y.cpitch + wbytes // 2 + _abytes(x) * (2 if acc else 1), flops, shp)]
    return _fp8_forward, _fp8_dgrad


ConvOp._fp8_forward, ConvOp._fp8_dgrad = _fp8_methods()


class BNMember:
    """one BatchNorm of a BNGroupOp (state only; the group emits the launches)"""

    def __init__(self, plan, bn, x, relu, residual, out):
        self.bn, self.x, self.relu, self.res, self.conv = bn, x, relu, residual, x.producer
        self.C, self.M = x.C, x.rows
        self.CP = cp8(self.C)
        self.y = out if out is not None else plan.act(x.N, x.T, x.H, x.W, x.C)
        # per-channel arrays are read with 16-byte loads: padded to CP
        self.mean, self.invstd, self.scale, self.shift = (plan.f32(self.CP) for _ in range(4))
        self.width = 2 * self.C + 1
        # y = relu(x*scale + shift) with nothing added: the backward recomputes the ReLU mask from x (which it reads for
        # xhat anyway) with the forward's expression and never touches y -- 5 tensor passes per BatchNorm instead of 7
        self.conv_bias = None        # Plan.bn(conv_bias=...)
        self.fused_pool = None       # the PoolOp that consumes y on the fly (Plan.maxpool(sole_consumer=True))
        self.fused_conv = None       # the ConvOp that applies this BatchNorm while it stages x (Plan.finalize, FUSE_BN_IN)
        self.reduce_fused = False    # the backward reduce runs in the epilogue of the consuming conv's data gradient
        self.apply_fused = False     # the backward apply runs inside the producing conv's weight gradient (dv_conv3d_wgrad_bn)
        self.mask_from_x = bool(relu) and residual is None
        self.sums_off = plan.reserve_zero(BN_REPLICAS * 2 * self.CP) if plan.with_grad else 0
        self.sums_len = BN_REPLICAS * 2 * self.CP
        # per-block partial sums + ticket of the ordered backward reduce (zero once: the kernel leaves the ticket zero)
        self.red_ws = (L.register_ticket_workspace(plan.f32(int(plan.lib.dv_bn_bwd_reduce_workspace(self.M, self.C)) // 4))
                       if plan.with_grad else None)


class BNGroupOp(Op):
    """Training-mode BatchNorm (+residual) (+ReLU) for one or several independent layers.

    With world > 1 the members' statistics travel in ONE all-gather forward and ONE all-reduce backward
    (SyncBatchNorm, torch/nn/modules/_functions.py:39-200): the four branch-entry BNs of an Inception block cost
    one small collective instead of four."""

    def __init__(self, plan, specs):
        super().__init__(plan)
        self.members = [BNMember(plan, *sp) for sp in specs]
        R = plan.comm.world
        self.width = sum(m.width for m in self.members)
        self.local = plan.f32(self.width)
        self.gathered = plan.f32(R, self.width) if plan.comm.exchange else self.local
        off = 0
        for m in self.members:
            m.loff = off
            off += m.width

    def grad_targets(self):
        if not self.plan.with_grad:
            return []
        return [('res%d' % i, m.res) for i, m in enumerate(self.members) if m.res is not None and m.res.grad is not None]

    def _multi(self, R, f_red, f_app, b_red, b_app):
        p, st, lib = self.plan, self.plan.store, self.plan.lib
        V = 4 if p.dtype == DV_F32 else 8
        arr = (L.BnItem * len(self.members))()
        ends = [0, 0, 0, 0]
        for i, m in enumerate(self.members):
            it, bn, x, y, res = arr[i], m.bn, m.x, m.y, m.res
            gs, bs = st.slot(bn.weight), st.slot(bn.bias)
            it.partials = m.conv.stats.data_ptr() + 4 * (x.off - m.conv.y.off) * m.conv.tiles      # [2][Cout][tiles]
            it.local_stats = self.local.data_ptr() + 4 * m.loff
            it.gamma, it.beta = st.w_master(gs), st.w_master(bs)
            it.running_mean = bn.running_mean.data_ptr() if bn.running_mean is not None else 0
            it.running_var = bn.running_var.data_ptr() if bn.running_var is not None else 0
            it.mean, it.invstd, it.scale, it.shift = (t.data_ptr() for t in (m.mean, m.invstd, m.scale, m.shift))
            it.x, it.ldx, it.y, it.ldy = x.ptr, x.ld, y.ptr, y.ld
            it.residual, it.ldr = (res.ptr, res.ld) if res is not None else (0, 0)
            it.M, it.C, it.n_tiles, it.tile_rows, it.pitch = m.M, m.C, m.conv.tiles, m.conv.tile_rows, m.conv.slot.Cout
            it.eps, it.momentum = float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.1)
            it.fwd_flags = DV_RELU if m.relu else 0
            total = m.M * (m.CP // V)
            ends[0] += m.C
            ends[1] += 0 if m.fused_conv is not None else max(1, min(4096, (total + 255) // 256))
            it.blk_stats, it.blk_apply = ends[0], ends[1]
            if p.with_grad:
                dres = res.grad if (res is not None and res.grad is not None) else None
                mflag = 0 if m.relu else DV_NO_RELU_MASK
                if m.mask_from_x:
                    mflag |= DV_MASK_FROM_X
                it.dy, it.lddy, it.dx, it.lddx = y.grad.ptr, y.grad.ld, x.grad.ptr, x.grad.ld
                it.dres, it.lddres = (dres.ptr, dres.ld) if dres is not None else (0, 0)
                it.sums, it.n_rep = p.zero_ptr(m.sums_off), BN_REPLICAS
                it.red_ws = m.red_ws.data_ptr()
                it.dgamma, it.dbeta = st.w_grad(gs), st.w_grad(bs)
                it.inv_count, it.dparam_scale = 1.0 / (m.M * R), 1.0 / R
                it.bwd_flags = mflag | (DV_ACCUM if (dres is not None and self.acc.get('res%d' % i)) else 0)
                ends[2] += 0 if m.reduce_fused else lib.dv_bn_bwd_blocks(m.M, m.C)
                ends[3] += 0 if m.apply_fused else max(1, min(2048, (total + 255) // 256))
                it.blk_red, it.blk_bapply = ends[2], ends[3]
        self._items = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(p.device)
        p.bytes += self._items.numel()
        tab, n, dt = self._items.data_ptr(), len(self.members), _dt(p.dtype)

        def tot(lst, attr):
            return sum(getattr(l, attr) for l in lst)
        f_red = [Launch('bn_stats_multi', 'bn_stats_multi', lib.dv_bn_stats_multi, (tab, n, 0 if p.comm.exchange else 1, ends[0]),
                        tot(f_red, 'bytes'))]
        f_app = [Launch('bn_apply_multi', 'bn_apply_multi<%s>' % dt, lib.dv_bn_apply_multi, (p.dtype, tab, n, ends[1]),
                        tot(f_app, 'bytes'))] if ends[1] else []
        # multi-rank step: ONE finalize launch for the group after the all-gather (instead of one per member)
        self._fin_multi = Launch('bn_finalize_multi', 'bn_finalize_multi', lib.dv_bn_finalize_multi,
                                 (tab, n, sum((m.C + 127) // 128 for m in self.members), self.local.data_ptr(),
                                  self.gathered.data_ptr(), R, self.width))
        if p.with_grad:
            b_red = [Launch('bn_bwd_reduce_multi', 'bn_bwd_reduce_multi<%s>' % dt, lib.dv_bn_bwd_reduce_multi,
                            (p.dtype, tab, n, ends[2]), tot(b_red, 'bytes'))] if ends[2] else []
            b_app = [Launch('bn_bwd_apply_multi', 'bn_bwd_apply_multi<%s>' % dt, lib.dv_bn_bwd_apply_multi,
                            (p.dtype, tab, n, ends[3], max(m.C for m in self.members)), tot(b_app, 'bytes'))] if ends[3] else []
            if b_app:
                b_app[0].gend = _gend(*[st.slot(t) for m in self.members for t in (m.bn.weight, m.bn.bias)])
        return f_red, f_app, b_red, b_app

    def _eval_launches(self):
        """module.eval(): y = act(x * scale + shift (+res)) with the affine map of the RUNNING statistics; no
        exchange, no update, no backward"""
        p, st, lib, dt = self.plan, self.plan.store, self.plan.lib, _dt(self.plan.dtype)
        f = []
        for m in self.members:
            bn, x, y, res = m.bn, m.x, m.y, m.res
            gs, bs = st.slot(bn.weight), st.slot(bn.bias)
            if bn.running_mean is None:
                raise NotImplementedError('eval-mode BatchNorm without running statistics')
            f.append(Launch('bn_eval_coeffs', 'bn_eval_coeffs', lib.dv_bn_eval_coeffs,
                            (st.w_master(gs), st.w_master(bs), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                             float(bn.eps), m.C, m.scale.data_ptr(), m.shift.data_ptr())))
            if m.conv_bias is not None:          # y = scale*(conv + b) + shift
                f.append(Launch('bn_bias_shift', 'addcmul', lib.dv_addcmul_f32,
                                (m.shift.data_ptr(), m.scale.data_ptr(), st.w_master(st.slot(m.conv_bias)), 1.0, m.C)))
            if m.fused_pool is not None:
                continue                     # applied by the pool op while it reads its windows
            f.append(Launch('bn_apply', 'bn_apply<%s>' % dt, lib.dv_bn_apply,
                            (p.dtype, x.ptr, x.ld, m.scale.data_ptr(), m.shift.data_ptr(),
                             res.ptr if res is not None else 0, res.ld if res is not None else 0, y.ptr, y.ld, m.M, m.C,
                             DV_RELU if m.relu else 0), _abytes(x) * (3 if res is not None else 2), 0, 'M%d C%d' % (m.M, m.C)))
        return f, []

    def launches(self):
        p, st, lib = self.plan, self.plan.store, self.plan.lib
        if not p.training:
            return self._eval_launches()
        R, dt = p.comm.world, _dt(p.dtype)
        f_red, f_fin, f_app, b_red, b_app = [], [], [], [], []
        for i, m in enumerate(self.members):
            bn, x, y, res, Cn, M = m.bn, m.x, m.y, m.res, m.C, m.M
            gs, bs = st.slot(bn.weight), st.slot(bn.bias)
            rm = bn.running_mean.data_ptr() if bn.running_mean is not None else 0
            rv = bn.running_var.data_ptr() if bn.running_var is not None else 0
            eps, mom = float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.1)
            local = self.local.data_ptr() + 4 * m.loff
            outs = (m.mean.data_ptr(), m.invstd.data_ptr(), m.scale.data_ptr(), m.shift.data_ptr())
            # x may be a channel slice of a merged conv's output: its partials are columns [coff, coff+C) of a wider table
            coff = x.off - m.conv.y.off
            spitch = m.conv.slot.Cout
            sptr = m.conv.stats.data_ptr() + 4 * coff * m.conv.tiles        # partials are [2][Cout][tiles]
            if not p.comm.exchange:
                f_red.append(Launch('bn_stats_finalize', 'bn_reduce_stats', lib.dv_bn_stats_finalize,
                                    (sptr, m.conv.tiles, m.conv.tile_rows, spitch, M, Cn, local, st.w_master(gs), st.w_master(bs),
                                     eps, mom, rm, rv) + outs, m.conv.tiles * 2 * Cn * 4))
            else:
                f_red.append(Launch('bn_reduce_stats', 'bn_reduce_stats', lib.dv_bn_reduce_stats,
                                    (sptr, m.conv.tiles, m.conv.tile_rows, spitch, M, Cn, local), m.conv.tiles * 2 * Cn * 4))
                f_fin.append(Launch('bn_finalize', 'bn_finalize', lib.dv_bn_finalize,
                                    (self.gathered.data_ptr() + 4 * m.loff, R, self.width, Cn, st.w_master(gs), st.w_master(bs),
                                     eps, mom, rm, rv) + outs))
            if m.fused_conv is None:         # (else: the consuming conv applies it on load; y is never written)
                f_app.append(Launch('bn_apply', 'bn_apply<%s>' % dt, lib.dv_bn_apply,
                                    (p.dtype, x.ptr, x.ld, m.scale.data_ptr(), m.shift.data_ptr(),
                                     res.ptr if res is not None else 0, res.ld if res is not None else 0, y.ptr, y.ld, M, Cn,
                                     DV_RELU if m.relu else 0), _abytes(x) * (3 if res is not None else 2), 0, 'M%d C%d' % (M, Cn)))
            if p.with_grad:
                dy = y.grad
                mflag = 0 if m.relu else DV_NO_RELU_MASK
                nact = 3 if (m.relu and not m.mask_from_x) else 2
                sums = p.zero_ptr(m.sums_off)
                if not m.reduce_fused:
                    b_red.append(Launch('bn_bwd_reduce', 'bn_bwd_reduce<%s>' % dt, lib.dv_bn_bwd_reduce,
                                        (p.dtype, dy.ptr, dy.ld, y.ptr, y.ld, x.ptr, x.ld, m.mean.data_ptr(), m.invstd.data_ptr(),
                                         M, Cn, mflag, sums, BN_REPLICAS, m.red_ws.data_ptr()), _abytes(x) * nact, 0,
                                        'M%d C%d' % (M, Cn)))
                dres = res.grad if (res is not None and res.grad is not None) else None
                bflags = mflag | (DV_ACCUM if (dres is not None and self.acc.get('res%d' % i)) else 0)
                nres = 0 if dres is None else (2 if bflags & DV_ACCUM else 1)
                if m.apply_fused:
                    continue
                b_app.append(Launch('bn_bwd_apply', 'bn_bwd_apply<%s>' % dt, lib.dv_bn_bwd_apply,
                                    (p.dtype, dy.ptr, dy.ld, y.ptr, y.ld, x.ptr, x.ld, m.mean.data_ptr(), m.invstd.data_ptr(),
                                     st.w_master(gs), sums, BN_REPLICAS, 1.0 / (M * R), 1.0 / R, st.w_grad(gs), st.w_grad(bs),
                                     x.grad.ptr, x.grad.ld, dres.ptr if dres is not None else 0,
                                     dres.ld if dres is not None else 0, M, Cn, bflags), _abytes(x) * (nact + 1 + nres), 0,
                                    'M%d C%d' % (M, Cn)))
                b_app[-1].gend = _gend(gs, bs)
        if len(self.members) > 1 or (p.with_grad and self.members[0].mask_from_x):
            # multi-tensor launches: one per phase for the whole group (the layers are small and latency bound); also
            # the form that carries scale / shift for the mask-from-x backward
            f1, a1 = f_red, f_app
            f_red, f_app, b_red, b_app = self._multi(R, f_red, f_app, b_red, b_app)
            if len(self.members) == 1:
                f_red, f_app = f1, a1        # a lone (large) layer keeps the single-tensor forward kernels (1024-thread stats)
            else:
                f_fin = [self._fin_multi]
        if self.members[0].fused_pool is not None:
            # y = relu(bn(x)) only feeds a max-pool: the pool op applies the BatchNorm while it reads its windows (its
            # backward writes dL/dy as before; the BatchNorm backward takes the ReLU mask from x, so y itself is never needed)
            f_app = []
        f = list(f_red)
        b = list(b_red)
        if p.comm.exchange:
            local, gathered, group = self.local, self.gathered, p.comm.group
            rc = p.comm.rccl
            if rc is not None:       # in-stream: ordered with the reduce kernels before and the finalize after, no hop
                f.append(Launch('syncbn_allgather', 'rccl:all_gather', rc.all_gather,
                                (local.data_ptr(), gathered.data_ptr(), self.width), 4 * self.width * (R + 1)))
            elif p.comm.flat_gather:
                f.append(HostStep('syncbn_allgather', lambda: dist.all_gather_into_tensor(gathered, local, group=group)))
            else:
                f.append(HostStep('syncbn_allgather', lambda: dist.all_gather(list(gathered.unbind(0)), local, group=group)))
            f += f_fin
            if p.with_grad:
                first, last = self.members[0], self.members[-1]
                lo, n = first.sums_off, last.sums_off + last.sums_len - first.sums_off
                pending = []

                # In place on the (contiguous) replica accumulators of the group, asynchronously: Plan.finalize moves
                # the weight-gradient kernels of the layers above between `start` and `wait`, so the exchange latency
                # (the dominant multi-GPU cost of this path) hides behind work that does not need it.
                def _start():
                    pending.append(dist.all_reduce(p.zero_arena.narrow(0, lo, n), group=group, async_op=True))

                def _wait():
                    pending.pop().wait()
                if rc is not None and os.environ.get('DUALVAR_BN_BWD_ASYNC') != '1':
                    # in-stream between the group's reduce and apply kernels.  Measured (one rank, every collective issued):
                    # the two event hops of the overlapped form cost ~24 us per exchange, more than a small RCCL
                    # all-reduce inside a node takes -- step 11.71 ms overlapped vs 10.95 ms in-stream (10.74 without exchange)
                    b.append(Launch('syncbn_allreduce', 'rccl:all_reduce', lambda stream: rc.all_reduce(p.zero_ptr(lo), n, stream), (),
                                    8 * n))
                elif rc is not None:
                    # same overlap with RCCL called directly: main --event--> exchange stream: all-reduce --event--> main
                    ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()
                    dev = p.device

                    def _start_direct(stream):
                        xs = p.comm.xstream(dev)
                        ev_a.record(torch.cuda.current_stream(dev))
                        xs.wait_event(ev_a)
                        L.check(rc.all_reduce(p.zero_ptr(lo), n, xs.cuda_stream), 'ncclAllReduce')
                        ev_b.record(xs)

                    def _wait_direct(stream):
                        torch.cuda.current_stream(dev).wait_event(ev_b)
                    b.append(StreamStep('syncbn_allreduce_start', _start_direct))
                    b.append(StreamStep('syncbn_allreduce_wait', _wait_direct))
                else:
                    b.append(HostStep('syncbn_allreduce_start', _start))
                    b.append(HostStep('syncbn_allreduce_wait', _wait))
        for m in self.members:               # conv bias in front of the BN: only the running mean sees it
            if m.conv_bias is not None and m.bn.running_mean is not None:
                assert len(self.members) == 1
                mom = float(m.bn.momentum if m.bn.momentum is not None else 0.1)
                f.append(Launch('bn_bias_running_mean', 'addcmul', lib.dv_addcmul_f32,
                                (m.bn.running_mean.data_ptr(), st.w_master(st.slot(m.conv_bias)), 0, mom, m.C)))
        f += f_app
        b += b_app
        return f, b


class PoolOp(Op):
    def __init__(self, plan, x, k, s, p, bn_member=None):
        super().__init__(plan)
        self.bn_member = bn_member
        self.x = x
        To, Ho, Wo = ops.conv_out_dims(x, k, s, p)
        self.y = plan.act(x.N, To, Ho, Wo, x.C)
        self.idx = torch.empty(self.y.rows, cp8(x.C), dtype=torch.uint8, device=plan.device)
        plan.bytes += self.idx.numel()
        self.d = ops.pool_desc(plan.dtype, x, self.y, k, s, p)
        self.need_dx = plan.with_grad and x.grad is not None
        if bn_member is not None:
            # forward fused with the BatchNorm + ReLU that produces x: the windows are read from the conv output and
            # normalised on the fly, x itself is never written (its gradient still is: the BatchNorm backward reads it)
            bn_member.fused_pool = self
            self.d_fused = ops.pool_desc(plan.dtype, bn_member.x, self.y, k, s, p)
            plan.bytes -= x.buf.numel() * x.buf.element_size()
            x.buf = x.buf.new_empty(0)

    def grad_targets(self):
        return [('x', self.x)] if self.need_dx else []

    def launches(self):
        p, lib, x, y = self.plan, self.plan.lib, self.x, self.y
        dt = _dt(p.dtype)
        if self.bn_member is not None:
            m = self.bn_member
            f = [Launch('bn_apply_maxpool', 'bn_apply_maxpool<%s>' % dt, lib.dv_bn_apply_maxpool,
                        (C.byref(self.d_fused), m.x.ptr, m.scale.data_ptr(), m.shift.data_ptr(), y.ptr, self.idx.data_ptr()),
                        _abytes(x) + _abytes(y) + y.rows * x.C)]
        else:
            f = [Launch('maxpool_fwd', 'maxpool_fwd<%s>' % dt, lib.dv_maxpool3d_fwd,
                        (C.byref(self.d), x.ptr, y.ptr, self.idx.data_ptr()), _abytes(x) + _abytes(y) + y.rows * x.C)]
        b = []
        if self.need_dx:
            acc = bool(self.acc.get('x'))
            b.append(Launch('maxpool_bwd', 'maxpool_bwd<%s>' % dt, lib.dv_maxpool3d_bwd,
                            (C.byref(self.d), y.grad.ptr, self.idx.data_ptr(), x.grad.ptr, DV_ACCUM if acc else 0),
                            _abytes(y) + y.rows * x.C + _abytes(x) * (2 if acc else 1)))
        return f, b


class GateGroupOp(Op):
    """S3D-G self gating of a whole Inception output, in place on the concat buffer (s3dg.py:68-78,112-128):
        cat[n,s,c] *= sigmoid(fc_i(mean_s cat[n,:,slice_i]))[c]
    Three launches forward (mean, the four FCs as ONE grouped fp32 MFMA GEMM with bias+sigmoid epilogue, scale) and
    four backward.  The un-gated activations are not kept: the gate is positive, so the ReLU mask of the producing
    BatchNorm is unchanged, and sum_s dy*y = (sum_s dy*out)/g."""

    def __init__(self, plan, fcs, cat):
        super().__init__(plan)
        self.fcs, self.cat = fcs, cat
        N, Ct = cat.N, cat.C
        assert sum(w for _, _, w in fcs) == Ct and all(o % 8 == 0 and w % 8 == 0 for _, o, w in fcs)
        self.mean, self.g = plan.f32(N, Ct), plan.f32(N, Ct)
        if plan.with_grad:
            self.dpre, self.dmean = plan.f32(N, Ct), plan.f32(N, Ct)

    def _table(self, descs):
        arr = (L.GemmDesc * len(descs))()
        tiles = 0
        for i, d in enumerate(descs):
            for k, v in d.items():
                setattr(arr[i], k, v)
            tiles += ((d['M'] + 31) // 32) * ((d['N'] + 31) // 32)
            arr[i].tile_end = tiles
        dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.plan.device)
        self.plan.bytes += dev.numel()
        return dev, tiles

    def launches(self):
        p, st, lib, cat = self.plan, self.plan.store, self.plan.lib, self.cat
        N, Ct, S, dt = cat.N, cat.C, cat.S, _dt(p.dtype)
        mean, g = self.mean.data_ptr(), self.g.data_ptr()
        fwd_d, bwd_d = [], []
        bias0 = None
        for fc, off, w in self.fcs:
            ws, bs = st.slot(fc.weight), st.slot(fc.bias)
            assert ws.cin_pitch == w
            if bias0 is None:
                bias0 = bs
            assert bs.off == bias0.off + off, 'gate biases must be registered back to back (see SepInception.register)'
            W = st.w_master(ws)
            fwd_d.append(dict(A=mean + 4 * off, sam=Ct, sak=1, B=W, sbk=1, sbn=w, C=g + 4 * off, ldc=Ct,
                              bias=st.w_master(bs), M=N, N=w, K=w, flags=DV_SIGMOID, alpha=1.0))
            if p.with_grad:
                dpre, dmean = self.dpre.data_ptr() + 4 * off, self.dmean.data_ptr() + 4 * off
                bwd_d.append(dict(A=dpre, sam=Ct, sak=1, B=W, sbk=w, sbn=1, C=dmean, ldc=Ct, bias=0,
                                  M=N, N=w, K=w, flags=0, alpha=1.0))                      # dmean = dpre @ W
                bwd_d.append(dict(A=dpre, sam=1, sak=Ct, B=mean + 4 * off, sbk=Ct, sbn=1, C=st.w_grad(ws), ldc=w, bias=0,
                                  M=w, N=w, K=N, flags=DV_ACCUM, alpha=1.0))               # dW += dpre^T @ mean
        self._ft, ftiles = self._table(fwd_d)
        f = [Launch('gate_mean', 'spatial_mean<%s>' % dt, lib.dv_spatial_mean,
                    (p.dtype, cat.ptr, cat.ld, N, S, Ct, mean), _abytes(cat)),
             Launch('gate_fc', 'gemm_f32_grouped', lib.dv_gemm_f32_grouped, (self._ft.data_ptr(), len(fwd_d), ftiles),
                    sum(w * w * 4 for _, _, w in self.fcs), sum(2 * N * w * w for _, _, w in self.fcs)),
             Launch('gate_scale', 'rowscale<%s,0>' % dt, lib.dv_gate_scale,
                    (p.dtype, cat.ptr, cat.ld, g, N, S, Ct, cat.ptr, cat.ld), 2 * _abytes(cat))]
        b = []
        if p.with_grad:
            dy = cat.grad
            self._bt, btiles = self._table(bwd_d)
            b = [Launch('gate_bwd_reduce', 'gate_bwd_reduce<%s>' % dt, lib.dv_gate_bwd_reduce,
                        (p.dtype, dy.ptr, dy.ld, cat.ptr, cat.ld, g, N, S, Ct, self.dpre.data_ptr(), 1), 2 * _abytes(cat)),
                 Launch('gate_fc_bwd', 'gemm_f32_grouped', lib.dv_gemm_f32_grouped, (self._bt.data_ptr(), len(bwd_d), btiles),
                        sum(w * w * 12 for _, _, w in self.fcs), sum(4 * N * w * w for _, _, w in self.fcs)),
                 Launch('gate_db', 'reduce_rows', lib.dv_colsum_f32, (self.dpre.data_ptr(), Ct, N, Ct, st.w_grad(bias0))),
                 Launch('gate_bwd_apply', 'rowscale<%s,1>' % dt, lib.dv_gate_bwd_apply,
                        (p.dtype, dy.ptr, dy.ld, g, self.dmean.data_ptr(), N, S, Ct, dy.ptr, dy.ld, 0), 2 * _abytes(cat))]
            b[1].gend = _gend(*[st.slot(fc.weight) for fc, _, _ in self.fcs])
            b[2].gend = _gend(*[st.slot(fc.bias) for fc, _, _ in self.fcs])
        return f, b


class MeanOp(Op):
    """global average pool -> fp32 [N, C] (AdaptiveAvgPool3d((1,1,1)))."""

    def __init__(self, plan, x):
        super().__init__(plan)
        self.x = x
        self.out = plan.f32(x.N, x.C)
        self.dout = plan.f32(x.N, x.C) if plan.with_grad else None

    def grad_targets(self):
        return [('x', self.x)] if (self.plan.with_grad and self.x.grad is not None) else []

    def launches(self):
        p, x, dt = self.plan, self.x, _dt(self.plan.dtype)
        f = [Launch('avgpool', 'spatial_mean<%s>' % dt, p.lib.dv_spatial_mean,
                    (p.dtype, x.ptr, x.ld, x.N, x.S, x.C, self.out.data_ptr()), _abytes(x))]
        b = []
        if p.with_grad and x.grad is not None:
            acc = bool(self.acc.get('x'))
            b = [Launch('avgpool_bwd', 'rowscale<%s,2>' % dt, p.lib.dv_spatial_mean_bwd,
                        (p.dtype, self.dout.data_ptr(), x.N, x.S, x.C, x.grad.ptr, x.grad.ld, DV_ACCUM if acc else 0),
                        _abytes(x) * (2 if acc else 1))]
        return f, b
