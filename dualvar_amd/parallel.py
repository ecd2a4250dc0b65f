"""Data parallelism for the pretrain step: one process per GPU, RCCL over xGMI.

What the reference gets from DistributedDataParallel + SyncBatchNorm + GatherLayer (pretrain.py:244-252,
SURVEY 2.2 C1-C6) is here:
  * C1 gradient averaging  -> `GradSync`: the whole encoder's gradient is ONE flat fp32 arena, so the
    all-reduce is a handful of large bucket collectives (default 32 MiB: ring all-reduce over xGMI is
    per-link bound, fewer/larger messages amortise the ~20-50 us launch+sync cost) issued on a side
    HIP stream; the 1/W averaging is folded into the SGD kernel's grad_scale.
  * C2 buffer broadcast    -> not needed: BN running stats come from global statistics and MoCo queues are
    filled from all-gathered keys, so they are identical on every rank by construction.
  * C3/C4 SyncBN stats     -> engine.BNOp (all_gather of (sum, M2, count) / all_reduce of the two backward sums)
  * C5/C6 feature gather   -> utils.GatherLayer
The host logic is device-agnostic so the world_size-2 tests run it on gloo/CPU tensors.
"""
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
    """callable(store) -> grad_scale.  All-reduces store.grad (sum) in buckets; returns 1/world."""

    def __init__(self, bucket_mb=32, group=None, side_stream=True):
        self.group = group
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.rank, self.world = world_info(group)
        self.side_stream = side_stream
        self._stream = None

    def reduce_flat(self, flat):
        if self.world == 1:
            return 1.0
        use_side = self.side_stream and flat.is_cuda
        if use_side:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            ctx = torch.cuda.stream(self._stream)
        else:
            ctx = _null()
        with ctx:
            works = [dist.all_reduce(flat[a:b], group=self.group, async_op=True)
                     for a, b in bucket_ranges(flat.numel(), self.bucket_elems)]
            for w in works:
                w.wait()
        if use_side:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        return 1.0 / self.world

    def __call__(self, store):
        return self.reduce_flat(store.grad)


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
