"""CPU, gloo, world_size 2: the host logic of the N>1 path -- GatherLayer semantics against the reference's
2-rank fixtures, bucketed gradient averaging, and the SyncBatchNorm statistic combine."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import gold


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dualvar_amd.parallel import GradSync, combine_bn_stats
        from dualvar_amd.utils.utils import concat_all_gather, gather_features
        from oracle import procedural as P, torch_ref as O
        out = {}
        # --- GatherLayer (product) + the oracle's loss arithmetic == the reference's 2-rank run
        N, B = 8, 8 // world
        clip = P.procedural_unit_features(N, 2, 128, seed=11)[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        allf = gather_features(clip, True)
        assert allf.shape == (N, 2, 128)
        f = allf.permute(1, 0, 2).reshape(2 * N, 128)
        logits = O.ntxent_from_similarity(f @ f.t(), torch.arange(2 * N), N, 0.07)
        loss = torch.nn.functional.cross_entropy(logits, torch.zeros(2 * N, dtype=torch.long))
        loss.backward()
        out['clip_logits'], out['clip_loss'], out['grad_clip'] = logits.detach().numpy(), float(loss), clip.grad.numpy()
        # --- bucketed all-reduce (sum) + 1/world scale == mean of the per-rank gradients
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        scale = GradSync(bucket_mb=0.001, side_stream=False).reduce_flat(flat)
        out['reduced'] = (flat * scale).numpy()
        # --- SyncBN: per-rank (sum, M2, count) gathered and combined == statistics of the whole batch
        x = torch.from_numpy(np.random.RandomState(3).standard_normal((40, 6))).float() * 2 + 1
        part = x[:15] if rank == 0 else x[15:]
        local = torch.cat([part.sum(0), ((part - part.mean(0)) ** 2).sum(0), torch.tensor([float(part.shape[0])])])
        gathered = concat_all_gather(local[None])
        mean, var = combine_bn_stats(gathered)
        out['bn_err'] = float(max((mean - x.mean(0)).abs().max(), (var - x.var(0, unbiased=False)).abs().max()))
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29611 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=300) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g = gold('losses')
    for r in range(2):
        o = res[r]
        assert np.allclose(o['clip_logits'], g[f'w2/r{r}/clip_logits'], atol=1e-5)
        assert abs(o['clip_loss'] - float(g[f'w2/r{r}/clip_contrast_loss'])) < 1e-6
        # GatherLayer.backward keeps this rank's slice of the FULL loss' gradient (utils.py:334-338)
        assert np.allclose(o['grad_clip'], g[f'w2/r{r}/grad_clip'], atol=1e-6)
        assert np.allclose(o['reduced'], np.arange(1000, dtype=np.float32) * 1.5)
        assert o['bn_err'] < 1e-5
