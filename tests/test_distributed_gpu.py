"""GPU: the data-parallel path end to end with 2 ranks sharing the one GPU of the test box (gloo backend, so no
RCCL "duplicate device" restriction; the collectives are the same torch.distributed calls RCCL serves on a node).

Identity checked: with SyncBatchNorm statistics and the GatherLayer feature exchange, the
clip loss on every rank equals the single-process loss on the concatenated batch, and the SUM over ranks of the
parameter gradients equals the single-process gradient (DDP then averages it: GradSync + SGD grad_scale)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

NET, B, T, H = 'r3d', 4, 8, 64


def _grads(model):
    return {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}


def _run(model, block):
    ret = model(block)
    loss = ret['clip_contrast_loss']
    for k in ret:
        if 'loss' in k and 'clip' not in k:
            loss = loss + ret[k]
    for st in model.stores():
        st.zero_grad()
    loss.backward()
    return {k: v.detach().float().cpu() for k, v in ret.items() if 'loss' in k or 'logits' in k}, _grads(model)


def _worker(rank, world, port, kind, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dualvar_amd import model as M
        from dualvar_amd.parallel import GradSync
        from oracle import procedural as P
        dev = torch.device('cuda:0')
        m = getattr(M, kind)(NET, 128, 0.07, True)
        P.procedural_init(m)
        m.set_compute_dtype('fp32').train().to(dev)
        V = 2 if kind.endswith('Naked') else 3
        full = P.procedural_clips(B, V, T, H, H)
        n = B // world
        np.random.seed(1234)
        if V == 3:          # every rank draws the permutations of ITS samples: replay the single-process stream
            perms = [np.random.permutation(2) for _ in range(B)]
            np.random.seed(1234)
            for _ in range(rank * n):
                np.random.permutation(2)
        outs, grads = _run(m, full[rank * n:(rank + 1) * n].to(dev))
        sync = GradSync(side_stream=False)
        scale = sync(m.store)
        torch.cuda.synchronize()
        synced = {k: (p.grad.detach().float().cpu() * scale).numpy() for k, p in m.named_parameters()}
        if True:
            # the same step with the all-reduce issued bucket by bucket from INSIDE the backward pass (GradSync.attach) --
            # for the dual-head objective from inside the LAST of its two encoder backward passes: the synchronised
            # gradient must not change (beyond the regrouping of the BatchNorm-backward sums)
            for side in (False, True):
                m.store.zero_grad()
                sync2 = GradSync(bucket_mb=1, side_stream=side)
                assert sync2.attach(m)
                np.random.seed(1234)
                if V == 3:
                    for _ in range(rank * n):
                        np.random.permutation(2)
                _run(m, full[rank * n:(rank + 1) * n].to(dev))
                early = len(sync2._works.get(id(m.store.grad), {}))
                scale2 = sync2(m.store)
                torch.cuda.synchronize()
                assert early >= 2, 'no bucket was reduced during the backward pass (%d)' % early
                for k, p in m.named_parameters():
                    got = (p.grad.detach().float().cpu() * scale2).numpy()
                    tol = 1e-4 * float(np.abs(synced[k]).max()) + 1e-9
                    assert float(np.abs(got - synced[k]).max()) <= tol, (side, k, float(np.abs(got - synced[k]).max()), tol)
                for mod in m.modules():
                    if hasattr(mod, 'grad_ready'):
                        mod.grad_ready = None
        q.put((rank, {k: v.numpy() for k, v in outs.items()}, {k: v.numpy() for k, v in grads.items()}, synced))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['SimCLR_Naked', 'SimCLR_TimeSeriesV4'])
def test_two_ranks_equal_one_process(gpu, kind):
    from dualvar_amd import model as M
    from oracle import procedural as P
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29711 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    [p.start() for p in procs]
    res = {}
    import queue
    import time
    deadline = time.time() + 300
    while len(res) < 2:
        try:
            r, outs, grads, synced = q.get(timeout=2)
        except queue.Empty:
            # a rank that died takes its peer's collectives with it: fail at once instead of waiting them out
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                [p.kill() for p in procs if p.is_alive()]
                pytest.fail('rank process died or timed out (exit codes %s)' % [p.exitcode for p in procs])
            continue
        t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}    # noqa: E731
        res[r] = (t(outs), t(grads), t(synced))
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)

    m = getattr(M, kind)(NET, 128, 0.07, False)
    P.procedural_init(m)
    m.set_compute_dtype('fp32').train().to(gpu)
    V = 2 if kind.endswith('Naked') else 3
    np.random.seed(1234)
    outs1, grads1 = _run(m, P.procedural_clips(B, V, T, H, H).to(gpu))

    for r in range(2):
        assert torch.allclose(res[r][0]['clip_logits'], outs1['clip_logits'], atol=2e-4)
        assert abs(float(res[r][0]['clip_contrast_loss']) - float(outs1['clip_contrast_loss'])) < 1e-5
    if V == 3:      # tc rows are per-rank: the mean over ranks is the single-process loss
        tc = np.mean([float(res[r][0]['tc_contrast_loss']) for r in range(2)])
        assert abs(tc - float(outs1['tc_contrast_loss'])) < 1e-5
    if kind == 'SimCLR_Naked':
        worst = 0.0
        for k, g1 in grads1.items():
            gsum = res[0][1][k] + res[1][1][k]
            e = float((gsum - g1).abs().max() / (g1.abs().max() + 1e-12))
            if e > worst:
                worst, worst_key = e, k
            # after GradSync every rank holds the same averaged gradient = single-process gradient / world
            assert torch.allclose(res[0][2][k], res[1][2][k])
            assert torch.allclose(res[0][2][k], gsum / 2, rtol=1e-5, atol=1e-8)
        # The per-rank problems are half as tall, so they may run with other tile shapes (= summation orders, and in fp32 mode
        # other groupings of the six partial products) than the single-process run.  Measured on MI355X, round 3, this seed:
        #   default (bf16-split products):  identity error 4.7e-6 (worst tensor: bn1.weight)
        #   DUALVAR_F32_EXACT=1:            identity error 7.0e-3 (worst: conv3.block1.conv1 weight gradient)
        # i.e. the split mode holds the identity THREE ORDERS tighter than the exact-f32 MFMA kernels: v_mfma_f32_32x32x2_f32
        # rounds every product into the running fp32 sum, so regrouping the rows of a long, cancelling weight-gradient sum between
        # the ranks moves it, while the bf16 partial products are exact in fp32 and enter the sum sixteen at a time.  The round-2
        # band of 3e-2 (sized for the exact kernels) was three orders of magnitude wider than what the product path delivers:
        # bound = 20x the measured identity error of the mode under test.  The 1e-7 input-nudge spread of the single-process
        # gradient is printed for the record (conditioning of the random-init R3D at B = 4).
        block = P.procedural_clips(B, V, T, H, H)
        noise = torch.from_numpy(np.random.RandomState(99).standard_normal(block.numel())).float().reshape(block.shape)
        np.random.seed(1234)
        _, grads_n = _run(m, (block * (1 + 1e-7 * noise)).to(gpu))
        spread = max(float((grads_n[k] - g1).abs().max() / (g1.abs().max() + 1e-12)) for k, g1 in grads1.items())
        np.random.seed(1234)
        _, grads_r = _run(m, block.to(gpu))
        assert all(torch.equal(grads_r[k], grads1[k]) for k in grads1), 'the single-process backward must be bit-reproducible'
        print('sum-over-ranks param grad vs single process: worst rel err %.3e (%s); single-process spread under a 1e-7 input nudge '
              '%.3e; F32_EXACT=%s' % (worst, worst_key, spread, os.environ.get('DUALVAR_F32_EXACT', '0')))
        from dualvar_amd import _lib
        assert worst < (1.5e-1 if _lib.f32_exact() else 1e-4), (worst, worst_key, spread)


# ---------------------------------------------------------------------------------------------------------------
# The reference's own 2-rank run of its three losses (tests/golden/losses.npz `w2/r{0,1}/*`, recorded by oracle/gen_golden.py
# from model/simclr.py under gloo): per-rank logits, losses and -- the part SURVEY 7 calls hard -- PER-RANK GRADIENTS
# (tc rows are this rank's slice only, GatherLayer.backward keeps only the own-rank gradient, utils/utils.py:334-338).
def _spawn2(target, args, timeout=300):
    import queue
    import time
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29511 + (os.getpid() % 200)
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + tuple(args)) for r in range(2)]
    [p.start() for p in procs]
    res, deadline = {}, time.time() + timeout
    while len(res) < 2:
        try:
            r, out = q.get(timeout=2)
            res[r] = out
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() > deadline:
                [p.kill() for p in procs if p.is_alive()]
                pytest.fail('rank process died or timed out (exit codes %s)' % [p.exitcode for p in procs])
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return res


def _loss_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import types
        from dualvar_amd import model as M
        from oracle import procedural as P
        dev = torch.device('cuda:0')
        N, B = 8, 8 // world
        mk = lambda *shape, seed: P.procedural_unit_features(N, *shape, seed=seed)[rank * B:(rank + 1) * B].clone().to(dev).requires_grad_(True)   # noqa: E731
        clip, ser, rk = mk(2, 128, seed=11), mk(2, 2, 64, seed=13), mk(2, 2, 64, seed=17)
        m = M.SimCLR_TimeSeriesV4.__new__(M.SimCLR_TimeSeriesV4)          # the loss methods only (no backbone)
        torch.nn.Module.__init__(m)
        m.distributed, m.T, m.aligned_T, m.n_series, m.series_dim, m.dim = True, 0.07, 0.07, 2, 64, 128
        m.args = types.SimpleNamespace(shufflerank_theta=0.05)
        r1 = m.calc_clip_contrast_loss(clip, 2)
        r2 = m.calc_tc_contrast_loss(ser)
        r3 = m.calc_ranking_loss(rk, 2, 'rank_', 0.5)
        (r1['clip_contrast_loss'] + r2['tc_contrast_loss'] + r3['rank_margin_contrast_loss']).backward()
        out = {}
        for r in (r1, r2, r3):
            for k, v in r.items():
                if 'rank0' not in k:
                    out[k] = v.detach().float().cpu().numpy()
        out['grad_clip'], out['grad_ser'], out['grad_rank'] = (t.grad.cpu().numpy() for t in (clip, ser, rk))
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_losses_against_reference_fixture(gpu):
    """dv_ntxent_fwd with a row offset (rank 1's tc rows start at row_index0 = n), dv_rank_margin and GatherLayer on the HIP
    path, two ranks: logits, losses and per-rank input gradients == the reference's 2-rank gloo run."""
    from tests.util import gold
    g = gold('losses')
    res = _spawn2(_loss_worker, ())
    for r in range(2):
        o = res[r]
        for k in ('clip_logits', 'tc_logits', 'rank_margin_logits'):
            ref = g[f'w2/r{r}/{k}']
            assert o[k].shape == ref.shape, (k, o[k].shape, ref.shape)
            assert float(np.abs(o[k] - ref).max()) < 2e-5, (r, k, float(np.abs(o[k] - ref).max()))
        for k in ('clip_contrast_loss', 'tc_contrast_loss', 'rank_margin_contrast_loss'):
            assert abs(float(o[k]) - float(g[f'w2/r{r}/{k}'])) < 2e-6, (r, k)
        for k in ('clip_labels', 'tc_labels', 'rank_margin_labels'):
            assert np.array_equal(o[k], g[f'w2/r{r}/{k}']), (r, k)
        for k in ('grad_clip', 'grad_ser', 'grad_rank'):
            ref = g[f'w2/r{r}/{k}']
            assert float(np.abs(o[k] - ref).max()) < 1e-6 + 1e-5 * float(np.abs(ref).max()), (r, k, float(np.abs(o[k] - ref).max()))
    # the per-rank tc gradients really differ from "half of the single-process gradient" (SURVEY 8c): the fixture is not trivial
    assert float(np.abs(g['w2/r0/tc_logits'] - g['w2/r1/tc_logits']).max()) > 1e-2


# ---------------------------------------------------------------------------------------------------------------
# MoCo under W = 2 (model/moco.py:110-127 _dequeue_and_enqueue all-gather, :129-173 batch shuffle, :337-355): the key
# all-gather fills identical queues on both ranks; with SyncBatchNorm statistics the two-rank step equals the single-process
# step on the concatenated batch (the product elides the clip shuffle: under global statistics it cannot change a result).
def _moco_worker(rank, world, port, q, kind):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dualvar_amd import model as M
        from dualvar_amd.optim import SGD
        from dualvar_amd.parallel import GradSync
        from oracle import procedural as P
        dev = torch.device('cuda:0')
        torch.manual_seed(0)
        m = getattr(M, kind)(NET, 128, 64, 0.999, 0.07, True)
        P.procedural_init(m)
        m.set_compute_dtype('fp32').train().to(dev)
        V = 2 if kind.endswith('Naked') else 3
        full = P.procedural_clips(B, V, T, H, H)
        n = B // world
        sync = GradSync(bucket_mb=1)
        sync.attach(m)
        opt = SGD([p for p in m.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4, stores=m.stores(),
                  grad_sync=sync)
        outs = []
        for it in range(2):
            np.random.seed(1234 + it)
            if V == 3:
                for _ in range(rank * n):
                    np.random.permutation(2)
            ret = m(full[rank * n:(rank + 1) * n].to(dev))
            loss = ret['clip_contrast_loss']
            for k in ret:
                if 'loss' in k and 'clip' not in k:
                    loss = loss + ret[k]
            opt.zero_grad()
            loss.backward()
            opt.step()
            outs.append({k: v.detach().float().cpu().numpy() for k, v in ret.items() if 'logits' in k or 'loss' in k})
        torch.cuda.synchronize()
        state = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()
                 if k in ('queue', 'series_queue', 'queue_ptr') or k.endswith('.2.weight')}
        q.put((rank, (outs, state)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['MoCo_Naked', 'MoCo_TimeSeriesV4'])
def test_moco_two_ranks_equal_one_process(gpu, kind):
    from dualvar_amd import model as M
    from dualvar_amd.optim import SGD
    from oracle import procedural as P
    res = _spawn2(_moco_worker, (kind,))
    torch.manual_seed(0)
    m = getattr(M, kind)(NET, 128, 64, 0.999, 0.07, False)
    P.procedural_init(m)
    m.set_compute_dtype('fp32').train().to(gpu)
    V = 2 if kind.endswith('Naked') else 3
    full = P.procedural_clips(B, V, T, H, H).to(gpu)
    opt = SGD([p for p in m.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4, stores=m.stores())
    single = []
    for it in range(2):
        np.random.seed(1234 + it)
        ret = m(full)
        loss = ret['clip_contrast_loss']
        for k in ret:
            if 'loss' in k and 'clip' not in k:
                loss = loss + ret[k]
        opt.zero_grad()
        loss.backward()
        opt.step()
        single.append({k: v.detach().float().cpu().numpy() for k, v in ret.items() if 'logits' in k or 'loss' in k})
    sd = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
    n = B // 2
    (o0, s0), (o1, s1) = res[0], res[1]
    # queues: identical on both ranks, bit for bit (filled from the same all-gathered keys), pointer advanced by B per step
    for k in s0:
        if 'queue' in k:
            assert np.array_equal(s0[k], s1[k]), k
    assert int(s0['queue_ptr'].reshape(-1)[0]) == int(sd['queue_ptr'].reshape(-1)[0]) == (2 * B) % 64
    # ... and equal to the single-process queue on the concatenated batch (SyncBN statistics are global)
    for k in ('queue', 'series_queue'):
        if k in s0:
            assert float(np.abs(s0[k] - sd[k]).max()) < 5e-5, (k, float(np.abs(s0[k] - sd[k]).max()))
    # step 0: a rank's logits are its rows of the single-process logits; the clip loss is the mean over ranks
    for r, o in ((0, o0), (1, o1)):
        assert float(np.abs(o[0]['clip_logits'] - single[0]['clip_logits'][r * n:(r + 1) * n]).max()) < 2e-4
    assert abs(0.5 * (float(o0[0]['clip_contrast_loss']) + float(o1[0]['clip_contrast_loss'])) - float(single[0]['clip_contrast_loss'])) < 1e-5
    # step 1 (after the averaged-gradient SGD step and the momentum update): still the single-process run
    tol = 5e-3
    for r, o in ((0, o0), (1, o1)):
        assert float(np.abs(o[1]['clip_logits'] - single[1]['clip_logits'][r * n:(r + 1) * n]).max()) < tol
    # the ranks hold identical weights after the step (same averaged gradient)
    for k in s0:
        if k.endswith('.2.weight'):
            assert np.array_equal(s0[k], s1[k]), k
            assert float(np.abs(s0[k] - sd[k]).max()) < 1e-4 * float(np.abs(sd[k]).max()) + 1e-7, k


# ---------------------------------------------------------------------------------------------------------------
# RCCL rehearsal: two ranks cannot share the one GPU of the test box under RCCL, so the backend="nccl" code path (flat
# all_gather_into_tensor, async all-reduce handles, collectives enqueued from the side stream) is run with ONE rank and
# DUALVAR_FORCE_EXCHANGE=1, which makes the engine and GradSync issue every collective of the multi-GPU step anyway.
# With one rank the exchanged statistics equal the local ones, so the step must reproduce the plain single-process step.
def _rccl_worker(port, kind, transport, net, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DUALVAR_FORCE_EXCHANGE='1', DUALVAR_RCCL=transport)
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        from dualvar_amd import model as M
        from dualvar_amd.optim import SGD
        from dualvar_amd.parallel import GradSync
        from oracle import procedural as P
        m = getattr(M, kind)(net, 128, 0.07, True)
        P.procedural_init(m)
        m.set_compute_dtype('fp32').train().to(dev)
        sync = GradSync(bucket_mb=1)
        assert sync.attach(m)
        opt = SGD([p for p in m.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4, stores=m.stores(),
                  grad_sync=sync)
        V = 2 if kind.endswith('Naked') else 3
        block = P.procedural_clips(B, V, T, H, H).to(dev)
        losses = []
        for _ in range(2):
            np.random.seed(1234)
            ret = m(block)
            loss = ret['clip_contrast_loss']
            for k in ret:
                if 'loss' in k and 'clip' not in k:
                    loss = loss + ret[k]
            opt.zero_grad()
            loss.backward()
            early = len(sync._works.get(id(m.store.grad), {}))
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        from dualvar_amd.engine import Comm
        assert Comm().exchange and Comm().flat_gather
        # 'direct': ncclAllGather / ncclAllReduce enqueued on the step's own streams (dualvar_amd/rccl.py); 'c10d': torch.distributed
        assert (Comm().rccl is not None) == (transport == 'direct') and (sync._rccl is not None) == (transport == 'direct')
        q.put((losses, early, {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items() if v.dtype.is_floating_point}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,transport,net', [('SimCLR_Naked', 'direct', 'r3d'), ('SimCLR_TimeSeriesV4', 'direct', 'r3d'),
                                                 ('SimCLR_Naked', 'c10d', 'r3d'), ('SimCLR_Naked', 'direct', 's3dg')])
def test_rccl_single_rank_rehearsal(gpu, kind, transport, net):
    from dualvar_amd import model as M
    from dualvar_amd.optim import SGD
    from oracle import procedural as P
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    proc = ctx.Process(target=_rccl_worker, args=(29911 + (os.getpid() % 200), kind, transport, net, q))
    proc.start()
    import queue
    import time
    deadline, got = time.time() + 300, None
    while got is None:
        try:
            got = q.get(timeout=2)
        except queue.Empty:
            if proc.exitcode not in (None, 0) or time.time() > deadline:
                if proc.is_alive():
                    proc.kill()
                pytest.fail('RCCL rehearsal process died or timed out (exit code %s)' % proc.exitcode)
    proc.join(timeout=120)
    assert proc.exitcode == 0
    losses, early, params = got
    assert early >= 2, 'no gradient bucket was all-reduced from inside the backward pass'

    V = 2 if kind.endswith('Naked') else 3

    def plain(perturb):
        """the same two steps without any exchange (plain single-process path)"""
        m = getattr(M, kind)(net, 128, 0.07, False)
        P.procedural_init(m)
        m.set_compute_dtype('fp32').train().to(gpu)
        opt = SGD([p for p in m.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4, stores=m.stores())
        block = P.procedural_clips(B, V, T, H, H)
        if perturb:
            noise = torch.from_numpy(np.random.RandomState(99).standard_normal(block.numel())).float().reshape(block.shape)
            block = block * (1 + perturb * noise)
        block = block.to(gpu)
        out = []
        for _ in range(2):
            np.random.seed(1234)
            ret = m(block)
            loss = ret['clip_contrast_loss']
            for k in ret:
                if 'loss' in k and 'clip' not in k:
                    loss = loss + ret[k]
            opt.zero_grad()
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        return out, {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()
                     if v.dtype.is_floating_point and 'num_batches' not in k}

    def spread(pa, pb):
        return max(float(np.abs(pa[k] - pb[k]).max() / (np.abs(pb[k]).max() + 1e-12)) for k in pb)

    want, ref = plain(0)
    # Two runs of the PLAIN path are bit-identical (no float atomics on the training path).  How far do they move under a
    # rounding-sized (1e-7) perturbation of the input?  The rehearsal takes other kernels through the statistics (reduce ->
    # gather -> finalize instead of one launch), i.e. other roundings: it cannot agree with the plain path better than the
    # plain path agrees with its own nudged self.
    _, again = plain(0)
    _, nudged = plain(1e-7)
    s_rr, s_in = spread(again, ref), spread(nudged, ref)
    assert s_rr == 0.0, s_rr
    print('RCCL single-rank rehearsal losses', losses, 'plain', want, '| plain-vs-plain spread: run-to-run %.2e, 1e-7 input nudge %.2e' % (s_rr, s_in))
    assert abs(losses[0] - want[0]) < 1e-5 * abs(want[0])
    assert abs(losses[1] - want[1]) < max(2e-3, 20 * max(s_rr, s_in)) * abs(want[1])     # after one SGD step
    worst = spread(params, ref)
    print('parameters after 2 steps: worst rel diff', worst)
    assert worst < max(2e-3, 20 * max(s_rr, s_in)), (worst, s_rr, s_in)


def test_bench_gpus_2_launches_itself_and_reports_two_ranks(gpu):
    """`python bench.py --gpus 2` with NO launcher on the command line (gloo rehearsal: both ranks share the one test GPU):
    the script starts its two ranks itself, every rank runs the data-parallel step, rank 0 prints ONE JSON line with
    n_gpus == 2 and the clips of both ranks in `value`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['DUALVAR_BENCH_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '4',
                        '--net', 'r3d', '--size', '64', '--secondary', 'none', '--no-cpu-baseline'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['clips_per_step'] == 2 * 4 * 2 and out['config']['parallelism'] == 'dp2'
    assert abs(out['value'] - out['config']['clips_per_step'] / (out['ms_per_step'] * 1e-3)) < 0.02 * out['value']
