"""Data parallelism for the pretrain step: one process per GPU, RCCL over xGMI.

What the reference gets from DistributedDataParallel + SyncBatchNorm + GatherLayer (pretrain.py:244-252,
SURVEY 2.2 C1-C6) is here:
  * C1 gradient averaging  -> `GradSync`: the whole encoder's gradient is ONE flat fp32 arena, so the
    all-reduce is a handful of large bucket collectives (default 8 MiB: ring all-reduce over xGMI is
    per-link bound, few large messages amortise the ~20-50 us launch+sync cost) issued on a side HIP
    stream, from inside the backward pass where the model allows it (GradSync.attach); the 1/W averaging
    is folded into the SGD kernel's grad_scale.
  * C2 buffer broadcast    -> not needed: BN running stats come from global statistics and MoCo queues are
    filled from all-gathered keys, so they are identical on every rank by construction.
  * C3/C4 SyncBN stats     -> engine.BNOp (all_gather of (sum, M2, count) / all_reduce of the two backward sums)
  * C5/C6 feature gather   -> utils.GatherLayer
The host logic is device-agnostic so the world_size-2 tests run it on gloo/CPU tensors.
"""
import os

import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def bucket_ranges(total, bucket_elems):
    """Contiguous [start, end) element ranges covering [0, total), last-to-first (backward completes the
    arena from its end, so the tail bucket is ready first)."""
    out = []
    end = total
    while end > 0:
        start = max(0, end - bucket_elems)
        out.append((start, end))
        end = start
    return out


class GradSync:
    """callable(store) -> grad_scale.  All-reduces store.grad (sum) in buckets; returns 1/world.

    `attach(model)` (optional) overlaps the reduction with the backward pass: the launch plan knows, for every point of
    its backward list, above which arena offset no gradient will be written any more (`engine.Plan._grad_triggers`), and
    calls `_on_ready` as each bucket -- last layers first, the heads' gradients are complete before the encoder's
    backward starts -- becomes final; that bucket's all-reduce then runs on the side stream underneath the rest of the
    backward.  Objectives that run the encoder several times per step (the TimeSeriesV4 ones accumulate all passes into
    the arena) arm this on the LAST encoder backward only (backbone/base.py: `pending_backward`).  The call from the
    optimizer reduces whatever is left and waits for everything."""

    def __init__(self, bucket_mb=8, group=None, side_stream=True):
        self.group = group
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.rank, self.world = world_info(group)
        # one rank + DUALVAR_FORCE_EXCHANGE=1: still issue the collectives (RCCL rehearsal on a single-GPU box, see engine.Comm)
        self.exchange = self.world > 1 or (os.environ.get('DUALVAR_FORCE_EXCHANGE') == '1' and dist.is_available()
                                           and dist.is_initialized())
        self.side_stream = side_stream
        self._stream = None
        self._works = {}             # id(flat) -> {bucket start: work handle} for the step in flight
        # a communicator of its own, called directly (dualvar_amd/rccl.py): the buckets do not queue behind the SyncBN
        # exchanges of c10d's / the engine's communicator.  None (gloo, DUALVAR_RCCL=c10d): torch.distributed calls.
        from . import rccl
        self._rccl = rccl.get('grad', group) if self.exchange else None

    # ---- overlap with backward
    def attach(self, model):
        if not self.exchange:
            return False
        for m in model.modules():
            if hasattr(m, '_plans') and hasattr(m, 'grad_ready'):
                m.grad_ready = self._on_ready
                m.bucket_elems = self.bucket_elems
        return True

    def _comm_stream(self, flat, *wait_for):
        if not (self.side_stream and flat.is_cuda):
            if flat.is_cuda:                 # collective on the current stream: it must still see the other streams' work
                for s_ in wait_for:
                    if s_ is not None:
                        torch.cuda.current_stream(flat.device).wait_stream(s_)
            return _null()
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=flat.device)
        self._stream.wait_stream(torch.cuda.current_stream(flat.device))
        for s_ in wait_for:
            if s_ is not None:
                self._stream.wait_stream(s_)
        return torch.cuda.stream(self._stream)

    def _all_reduce(self, flat, a, b):
        """sum-all-reduce flat[a:b] on the current stream; -> work handle (torch.distributed) or None (enqueued in-stream)"""
        if self._rccl is not None and flat.is_cuda:
            rc = self._rccl.all_reduce(flat.data_ptr() + 4 * a, b - a, torch.cuda.current_stream(flat.device).cuda_stream)
            if rc:
                raise RuntimeError('ncclAllReduce failed with %d' % rc)
            return None
        return dist.all_reduce(flat[a:b], group=self.group, async_op=True)

    def _on_ready(self, plan, lo):
        """gradient-arena elements [lo, total) are final: start the buckets that lie in there"""
        flat = plan.store.grad
        works = self._works.setdefault(id(flat), {})
        todo = [(a, b) for a, b in bucket_ranges(flat.numel(), self.bucket_elems) if a >= lo and a not in works]
        if not todo:
            return
        plan.store._sync_started = True          # from here on nothing may write this arena before the optimizer step
        plan.store._sync_owner = self             # (ParamStore.zero_grad abandons a step whose optimizer step never came)
        with self._comm_stream(flat, getattr(plan, '_side', None)):
            for a, b in todo:
                works[a] = self._all_reduce(flat, a, b)

    # ---- the reduction proper (optimizer step)
    def reduce_flat(self, flat):
        if not self.exchange:
            return 1.0
        works = self._works.pop(id(flat), {})
        with self._comm_stream(flat):
            for a, b in bucket_ranges(flat.numel(), self.bucket_elems):
                if a not in works:
                    works[a] = self._all_reduce(flat, a, b)
            for w in works.values():
                if w is not None:
                    w.wait()
        if self.side_stream and flat.is_cuda:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        return 1.0 / self.world

    def __call__(self, store):
        scale = self.reduce_flat(store.grad)
        store._sync_started = False
        return scale

    def abandon(self, store):
        """A step whose bucket all-reduces were started from inside the backward but whose optimizer step never came (an
        exception, a skipped / NaN step, zero_grad() and retry): wait for what is in flight -- it writes the arena -- forget it
        and re-arm the arena for the next backward.  Every rank must take the same decision, as with any collective."""
        flat = store.grad
        works = self._works.pop(id(flat), {}) if flat is not None else {}
        for w in works.values():
            if w is not None:
                w.wait()
        if works and self._stream is not None and flat is not None and flat.is_cuda:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        store._sync_started = False


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def combine_bn_stats(stats):
    """Reference (host) formula of dv_bn_finalize's cross-rank combine, for the CPU tests:
    stats [R, 2C+1] rows of (sum[C], M2[C], count) -> (mean[C], biased var[C])."""
    C = (stats.shape[1] - 1) // 2
    cnt = stats[:, 2 * C]
    tot = cnt.sum()
    mean = stats[:, :C].sum(0) / tot
    m_r = stats[:, :C] / cnt[:, None]
    M2 = (stats[:, C:2 * C] + cnt[:, None] * (m_r - mean) ** 2).sum(0)
    return mean, M2 / tot
