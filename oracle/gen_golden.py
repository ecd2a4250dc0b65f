"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz from the REFERENCE's own classes.

Run in the build container only (needs /root/reference):   python -m oracle.gen_golden
For every case it (1) runs the reference (imported through oracle/harness.py), (2) runs
the oracle restatement (oracle/torch_ref.py) on the same procedural weights and inputs,
(3) asserts they agree to 1e-5, and (4) stores the *reference's* outputs.  Only data is
written (inputs are procedural, so fixtures hold outputs and the numpy permutation).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import harness, procedural as P, torch_ref as O   # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
ARGS = types.SimpleNamespace(shufflerank_theta=0.05)
CLIP = dict(T=8, H=112, W=112)
WC_LR = 3e-7         # learning rate of the well-conditioned multi-step runs (case_models)


def _init_pg(rank=0, world=1, port=29531):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if not dist.is_initialized():
        dist.init_process_group('gloo', rank=rank, world_size=world)


def build(ns, kind, net, distributed, K=64):
    S, M = (ns.simclr, ns.moco) if hasattr(ns, 'simclr') else (ns, ns)
    if kind == 'simclr_naked':
        return S.SimCLR_Naked(net, 128, 0.07, distributed)
    if kind == 'simclr_timeseriesv4':
        return S.SimCLR_TimeSeriesV4(net, 128, 0.07, distributed, args=ARGS)
    if kind == 'moco_naked':
        return M.MoCo_Naked(net, 128, K, 0.999, 0.07, distributed)
    if kind == 'moco_timeseriesv4':
        return M.MoCo_TimeSeriesV4(net, 128, K, 0.999, 0.07, distributed, args=ARGS)
    raise KeyError(kind)


def grad_summary(model):
    """{canonical key: [sum|g|, sum g]} for every trainable tensor (compact gradient pin)."""
    sd = model.state_dict(keep_vars=True)
    out = {}
    for key in sorted(P.canonical_groups(model)):
        t = sd[key]
        if getattr(t, 'grad', None) is not None:
            g = t.grad.double()
            out[key] = np.array([g.abs().sum().item(), g.sum().item()])
    return out


def param_checksum(model):
    sd = model.state_dict()
    return {k: np.array([sd[k].double().abs().sum().item(), sd[k].double().sum().item()])
            for k in sorted(P.canonical_groups(model)) if sd[k].dtype.is_floating_point}


def run_model(model, block, steps, np_seed, lr=0.003):
    """`steps` iterations of pretrain.py:394-451 on the same block.  Records step-0 outputs,
    step-0 gradients (checksums of every tensor + element-wise samples of ~30 of them, oracle/procedural.py:
    grad_samples), and the parameter checksum after the last step."""
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD([{'params': [p]} for p in params], lr=lr, weight_decay=1e-4, momentum=0.9)
    np.random.seed(np_seed)
    rec = {}
    for it in range(steps):
        state = np.random.get_state()
        ret = model(block)
        after = np.random.get_state()
        loss = 0
        if 'clip_contrast_loss' in ret:
            loss = ret['clip_contrast_loss']
        for key in ret:
            if 'loss' in key and 'clip' not in key:
                loss = loss + ret[key]
        opt.zero_grad()
        loss.backward()
        if it in (0, steps - 1):
            tag = 'first' if it == 0 else 'last'
            if block.shape[1] == 3:                      # the permutation the forward just drew
                np.random.set_state(state)
                rec[f'{tag}/perm'] = np.array([np.random.permutation(2) for _ in range(block.shape[0])])
                np.random.set_state(after)
            for k, v in ret.items():
                rec[f'{tag}/out/' + k] = v.detach().numpy().copy()
            rec[f'{tag}/total_loss'] = np.array(float(loss))
            for k, v in grad_summary(model).items():
                rec[f'{tag}/grad/' + k] = v
            if it == 0:
                for k, v in P.grad_samples(model).items():
                    rec['first/gsample/' + k] = v
        opt.step()
        rec['loss_step%d' % it] = np.array(float(loss))
    for k, v in param_checksum(model).items():
        rec['param/' + k] = v
    if hasattr(model, 'queue_ptr'):
        rec['queue_ptr'] = model.queue_ptr.numpy().copy()
        rec['queue_cols'] = model.queue[:, :16].numpy().copy()
    return rec


def gsample_noise_floor(ns, kind, net, distributed, block, np_seed=1234):
    """max |g_fp32 - g_fp64| per sampled gradient tensor, for the REFERENCE itself at step 0: what rounding alone does to a
    correct fp32 implementation (sums over 1e5 rows with cancellation -- BatchNorm beta / gamma gradients -- sit far above
    `1e-4 max|g|`).  Tests never ask for better agreement than a multiple of this."""
    outs = []
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(0)
        m = build(ns, kind, net, distributed)
        P.procedural_init(m)
        m.train().to(dt)
        np.random.seed(np_seed)
        ret = m(block.to(dt))
        loss = ret['clip_contrast_loss'] if 'clip_contrast_loss' in ret else 0
        for key in ret:
            if 'loss' in key and 'clip' not in key:
                loss = loss + ret[key]
        loss.backward()
        outs.append(P.grad_samples(m))
    return {k: float(np.max(np.abs(outs[0][k] - outs[1][k]))) for k in outs[0]}


def step_loss_noise_floor(ns, kind, net, distributed, block, steps, np_seed=1234, lr=0.003):
    """|loss_fp32 - loss_fp64| of the REFERENCE itself after each SGD step of run_model's loop: what fp32 rounding ANYWHERE in
    the network (not only a perturbed input, `sens/*`) does to a correct implementation's step losses.  The summation order of a
    from-scratch kernel differs from ATen's in every layer, so tests never ask a post-step loss for better agreement than a
    multiple of this (round 4: the LDS-staged conv kernel sums channel chunks outside taps; R(2+1)D's loss after one step at
    lr = 0.003 moved from 1.1e-3 to 1.3e-3 of the reference's with errors against float64 unchanged)."""
    losses = []
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(0)
        m = build(ns, kind, net, distributed)
        P.procedural_init(m)
        m.train().to(dt)
        params = [p for p in m.parameters() if p.requires_grad]
        opt = torch.optim.SGD([{'params': [p]} for p in params], lr=lr, weight_decay=1e-4, momentum=0.9)
        np.random.seed(np_seed)
        cur = []
        for it in range(steps):
            ret = m(block.to(dt))
            loss = ret['clip_contrast_loss'] if 'clip_contrast_loss' in ret else 0
            for key in ret:
                if 'loss' in key and 'clip' not in key:
                    loss = loss + ret[key]
            opt.zero_grad()
            loss.backward()
            opt.step()
            cur.append(float(loss))
        losses.append(cur)
    return {'f64/loss_step%d' % it: abs(losses[0][it] - losses[1][it]) for it in range(steps)}, losses[0]


def case_models_add_f64_step_losses(ref):
    """adds the `f64/loss_step*` entries to the committed model fixtures WITHOUT regenerating them (the fp32 leg must reproduce
    the stored step losses bit for bit, or this refuses)"""
    _init_pg()
    for kind, net, B in (('simclr_naked', 's3dg', 4), ('simclr_timeseriesv4', 's3dg', 4),
                         ('simclr_timeseriesv4', 'r21d', 2), ('simclr_naked', 'r3d', 2),
                         ('moco_naked', 's3dg', 4), ('moco_timeseriesv4', 's3dg', 4)):
        path = os.path.join(GOLD, f'model_{kind}_{net}.npz')
        old = dict(np.load(path))
        V = 2 if kind.endswith('naked') else 3
        block = P.procedural_clips(B, V, **CLIP)
        steps = {'moco_naked': 3, 'moco_timeseriesv4': 1}.get(kind, 2)
        add, l32 = step_loss_noise_floor(ref, kind, net, True, block, steps)
        for it in range(steps):
            assert l32[it] == float(old['loss_step%d' % it]), (kind, net, it, l32[it], float(old['loss_step%d' % it]))
        for k, v in add.items():
            old[k] = np.array(v)
        np.savez_compressed(path, **old)
        print('model', kind, net, 'fp32-vs-fp64 step losses of the reference:', {k: '%.2e' % v for k, v in add.items()},
              'sens', {('loss_step%d' % it): '%.2e' % float(old['sens/loss_step%d' % it]) for it in range(steps)})


def compare(a, b, tag, tol=5e-4):
    a = {k: v for k, v in a.items() if not k.startswith(('sens/', 'f64/'))}
    b = {k: v for k, v in b.items() if not k.startswith(('sens/', 'f64/'))}
    assert a.keys() == b.keys(), (tag, set(a) ^ set(b))
    worst = 0.0
    for k in a:
        x, y = np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64)
        assert x.shape == y.shape, (tag, k, x.shape, y.shape)
        err = float(np.max(np.abs(x - y) / (1.0 + np.abs(x)))) if x.size else 0.0
        worst = max(worst, err)
        # later steps amplify last-bit differences of step 0 (the net is chaotic at B=4): looser there
        assert err < (tol if k.startswith('first/') else 20 * tol), (tag, k, err)
    return worst


def case_backbones(ref):
    rec = {}
    for net in ('s3dg', 'r21d', 'r3d', 'r50'):
        x = P.procedural_clips(4, 1, **CLIP)[:, 0]
        outs = []
        for sel in (ref.select_backbone, O.select_backbone):
            m, _ = sel(net)
            P.procedural_init(m).train()
            y = m(x)
            outs.append(y)
        err = float((outs[0] - outs[1]).abs().max())
        assert err < 1e-5, (net, err)
        rec[net + '/feat'] = outs[0].detach().numpy()
        rec[net + '/pooled'] = outs[0].mean(dim=(2, 3, 4)).detach().numpy()
        # conditioning of the case: the reference's own fp32 result vs the same network in fp64
        with torch.no_grad():
            y64 = m.double()(x.double())
        cond = float((outs[0].double() - y64).abs().max() / y64.abs().max())
        rec[net + '/fp32_vs_fp64'] = np.array(cond)
        print('backbone', net, tuple(outs[0].shape), 'ref-vs-oracle', err, 'reference fp32-vs-fp64', cond)
    np.savez_compressed(os.path.join(GOLD, 'backbones.npz'), **rec)


def case_eval(ref):
    """module.eval() (classifier.py's test / retrieval passes): every backbone after ONE train-mode forward (so the
    running statistics are not the trivial 0 / 1) evaluated on other clips, and the LinearClassifier around it."""
    rec = {}
    xa = P.procedural_clips(4, 1, **CLIP)[:, 0]
    xb = P.procedural_clips(4, 1, seed=77, **CLIP)[:, 0]
    for net in ('s3dg', 'r21d', 'r3d', 'r50'):
        outs = []
        for sel in (ref.select_backbone, O.select_backbone):
            m, _ = sel(net)
            P.procedural_init(m).train()
            with torch.no_grad():
                m(xa)
                y = m.eval()(xb)
            outs.append(y)
        err = float((outs[0] - outs[1]).abs().max())
        assert err < 1e-5, (net, err)
        rec[net + '/eval_pooled'] = outs[0].mean(dim=(2, 3, 4)).numpy()
        print('eval backbone', net, 'ref-vs-oracle', err)
    for tag, kw in (('plain', dict(use_dropout=True)), ('l2bn', dict(use_dropout=False, use_l2_norm=True, use_final_bn=True)),
                    ('mlp', dict(use_dropout=False, nonlinear=True))):
        outs = []
        for mk in (lambda **k: ref.linear_classifier('s3dg', **k), lambda **k: O.LinearClassifier(network='s3dg', **k)):
            torch.manual_seed(0)
            c = mk(num_class=101, **kw)
            P.procedural_init(c).train()
            with torch.no_grad():
                c.backbone(xa)
                logit, feat = c.eval()(xb)
            outs.append((logit, feat))
        err = max(float((outs[0][i] - outs[1][i]).abs().max()) for i in range(2))
        assert err < 1e-5, (tag, err)
        rec['clf_%s/logit' % tag], rec['clf_%s/feat' % tag] = outs[0][0].numpy(), outs[0][1].numpy()
        print('eval classifier', tag, 'ref-vs-oracle', err)
    np.savez_compressed(os.path.join(GOLD, 'eval.npz'), **rec)


def _clf_train(c, mode, xa, xb, labels, steps=2, lr=0.01):
    """classifier.py:240-262,422-470 restated: 'ft' = model.train(), every parameter trained; 'last' = model.eval(),
    final_bn.train(), backbone frozen.  SGD(momentum 0.9, wd 1e-4); CrossEntropyLoss."""
    with torch.no_grad():
        c.train()
        c.backbone(xa)                                   # non-trivial running statistics
    if mode == 'last':
        for n_, p_ in c.named_parameters():
            if 'backbone' in n_:
                p_.requires_grad = False
    params = [{'params': [p_]} for p_ in c.parameters() if p_.requires_grad]
    opt = torch.optim.SGD(params, lr=lr, momentum=0.9, weight_decay=1e-4)
    crit = torch.nn.CrossEntropyLoss()
    rec = {}
    for it in range(steps):
        if mode == 'last':
            c.eval()
            if getattr(c, 'use_final_bn', False):
                c.final_bn.train()
        else:
            c.train()
        logit, feat = c(xb)
        loss = crit(logit, labels)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            rec['logit0'], rec['feat0'] = logit.detach().numpy().copy(), feat.detach().numpy().copy()
            for k, v in grad_summary(c).items():
                rec['grad/' + k] = v
        opt.step()
        rec['loss%d' % it] = np.array(float(loss))
    for k, v in param_checksum(c).items():
        rec['param/' + k] = v
    with torch.no_grad():
        rec['eval_logit'] = c.eval()(xb)[0].numpy().copy()
    return rec


def case_classifier_train(ref):
    """downstream finetune steps on r3d, B = 4: 'ft' without dropout (its mask cannot be reproduced
    across devices) and 'last' with L2 norm + final BatchNorm1d (dropout is off in that mode anyway)"""
    xa = P.procedural_clips(4, 1, **CLIP)[:, 0]
    xb = P.procedural_clips(4, 1, seed=77, **CLIP)[:, 0]
    labels = torch.tensor([3, 0, 2, 1])
    out = {}
    for mode, kw in (('ft', dict(use_dropout=False)), ('last', dict(use_dropout=True, use_l2_norm=True, use_final_bn=True))):
        recs = []
        for mk in (lambda **k: ref.linear_classifier('r3d', **k), lambda **k: O.LinearClassifier(network='r3d', **k)):
            torch.manual_seed(0)
            c = mk(num_class=10, **kw)
            P.procedural_init(c)
            recs.append(_clf_train(c, mode, xa, xb, labels))
        err = compare(recs[0], recs[1], ('clf', mode))
        # sensitivity: the reference re-run on a 1e-6-perturbed input
        torch.manual_seed(0)
        c = ref.linear_classifier('r3d', num_class=10, **kw)
        P.procedural_init(c)
        noise = torch.from_numpy(np.random.RandomState(99).standard_normal(xb.numel())).float().reshape(xb.shape)
        pert = _clf_train(c, mode, xa, xb * (1 + 1e-6 * noise), labels)
        for k in list(recs[0].keys()):
            a, b = np.asarray(recs[0][k], dtype=np.float64), np.asarray(pert[k], dtype=np.float64)
            recs[0]['sens/' + k] = np.array(float(np.max(np.abs(a - b))))
        for k, v in recs[0].items():
            out[mode + '/' + k] = v
        print('classifier train', mode, 'ref-vs-oracle', err, 'loss', float(recs[0]['loss0']), float(recs[0]['loss1']))
    np.savez_compressed(os.path.join(GOLD, 'classifier_train.npz'), **out)


def case_backbones_extra(ref):
    """the remaining names of the reference factory (select_backbone.py:9-27): r2d3d18, c3d and s3d (S3D without
    self-gating, backbone/s3dg.py:135 gating=False) -- train-mode features,
    then eval() on other clips (pins the conv-bias handling of c3d through the running mean)"""
    rec = {}
    xa = P.procedural_clips(4, 1, **CLIP)[:, 0]
    xb = P.procedural_clips(4, 1, seed=77, **CLIP)[:, 0]
    for net in ('r2d3d18', 'c3d', 's3d'):
        outs = []
        for sel in (ref.select_backbone, O.select_backbone):
            m, _ = sel(net)
            P.procedural_init(m).train()
            with torch.no_grad():
                y = m(xa)
                ye = m.eval()(xb)
            outs.append((y, ye, m))
        err = max(float((outs[0][i] - outs[1][i]).abs().max()) for i in range(2))
        assert err < 1e-5, (net, err)
        rec[net + '/feat'] = outs[0][0].numpy()
        rec[net + '/pooled'] = outs[0][0].mean(dim=(2, 3, 4)).numpy()
        rec[net + '/eval_pooled'] = outs[0][1].mean(dim=(2, 3, 4)).numpy()
        m = outs[0][2]
        with torch.no_grad():
            m2, _ = ref.select_backbone(net)
            P.procedural_init(m2).train()
            y64 = m2.double()(xa.double())
        rec[net + '/fp32_vs_fp64'] = np.array(float((outs[0][0].double() - y64).abs().max() / y64.abs().max()))
        print('backbone', net, tuple(outs[0][0].shape), 'ref-vs-oracle', err, 'fp32-vs-fp64', float(rec[net + '/fp32_vs_fp64']))
    np.savez_compressed(os.path.join(GOLD, 'backbones_extra.npz'), **rec)


def case_models(ref):
    _init_pg()
    only = os.environ.get('GOLDEN_ONLY')                 # e.g. moco_timeseriesv4:s3dg -- regenerate one fixture
    for kind, net, B in (('simclr_naked', 's3dg', 4), ('simclr_timeseriesv4', 's3dg', 4),
                         ('simclr_timeseriesv4', 'r21d', 2), ('simclr_naked', 'r3d', 2),
                         ('moco_naked', 's3dg', 4), ('moco_timeseriesv4', 's3dg', 4)):
        if only and only != f'{kind}:{net}':
            continue
        V = 2 if kind.endswith('naked') else 3
        block = P.procedural_clips(B, V, **CLIP)
        # SimCLR_Naked in the reference only works on the distributed path (D2): world_size 1 gloo
        distributed = True
        recs = []
        for ns in (ref, O):
            torch.manual_seed(0)
            m = build(ns, kind, net, distributed)
            P.procedural_init(m)
            recs.append(run_model(m, block, steps={'moco_naked': 3, 'moco_timeseriesv4': 1}.get(kind, 2), np_seed=1234))
        err = compare(recs[0], recs[1], (kind, net))
        # sensitivity of the case: the REFERENCE re-run on an input perturbed by 1e-6 (relative).  Tests use it
        # to scale their tolerances: a from-scratch implementation cannot agree with the reference better than
        # the reference agrees with itself under rounding-sized perturbations.
        torch.manual_seed(0)
        m = build(ref, kind, net, distributed)
        P.procedural_init(m)
        noise = torch.from_numpy(np.random.RandomState(99).standard_normal(block.numel())).float().reshape(block.shape)
        pert = run_model(m, block * (1 + 1e-6 * noise), steps={'moco_naked': 3, 'moco_timeseriesv4': 1}.get(kind, 2), np_seed=1234)
        for k in list(recs[0].keys()):
            if k in pert and ('/out/' in k or k.startswith('loss_step') or k.startswith('param/') or '/grad/' in k
                              or '/gsample/' in k) and 'labels' not in k:
                a, b = np.asarray(recs[0][k], dtype=np.float64), np.asarray(pert[k], dtype=np.float64)
                recs[0]['sens/' + k] = np.array(float(np.max(np.abs(a - b))))
        for k, v in gsample_noise_floor(ref, kind, net, distributed, block).items():
            recs[0]['f64/first/gsample/' + k] = np.array(v)
        for k, v in step_loss_noise_floor(ref, kind, net, distributed, block, {'moco_naked': 3, 'moco_timeseriesv4': 1}.get(kind, 2))[0].items():
            recs[0][k] = np.array(v)
        # the oracle's non-distributed path must equal the distributed one at world_size 1
        torch.manual_seed(0)
        m = build(O, kind, net, False)
        P.procedural_init(m)
        err2 = compare(recs[0], run_model(m, block, steps={'moco_naked': 3, 'moco_timeseriesv4': 1}.get(kind, 2), np_seed=1234),
                       (kind, net, 'nondist'))
        if net == 's3dg':
            # WELL-CONDITIONED multi-step run: at this initialisation the gradients of the BatchNorm-ed convs are so large
            # that the paper's lr = 0.003 is a leap into another basin -- after ONE such step a 1e-6 input perturbation
            # moves the reference's own logits by 1.3, so `last/*` above cannot pin the backward + optimizer.  With
            # lr = 3e-7 three SGD steps (momentum engaged) still move the loss by ~0.3 while the reference's own
            # sensitivity stays ~1e-4 on the loss and ~3e-3 on the logits: `wc/*` pins them to a fraction of a per cent.
            wc = []
            for ns in (ref, O):
                torch.manual_seed(0)
                m = build(ns, kind, net, distributed)
                P.procedural_init(m)
                wc.append(run_model(m, block, steps=3, np_seed=1234, lr=WC_LR))
            keep = lambda r: {k: v for k, v in r.items() if k.startswith(('loss_step', 'last/out/', 'param/', 'queue_ptr'))}   # noqa: E731
            wc = [keep(r) for r in wc]
            errw = compare(wc[0], wc[1], (kind, net, 'wc'), tol=5e-3 if kind == 'moco_timeseriesv4' else 5e-4)
            torch.manual_seed(0)
            m = build(ref, kind, net, distributed)
            P.procedural_init(m)
            pertw = run_model(m, block * (1 + 1e-6 * noise), steps=3, np_seed=1234, lr=WC_LR)
            pertw = keep(pertw)
            for k, v in wc[0].items():
                recs[0]['wc/' + k] = v
                if k in pertw and ('/out/' in k or k.startswith('loss_step') or k.startswith('param/')) and 'labels' not in k:
                    a, b = np.asarray(v, dtype=np.float64), np.asarray(pertw[k], dtype=np.float64)
                    recs[0]['wc/sens/' + k] = np.array(float(np.max(np.abs(a - b))))
            print('   well-conditioned run (lr %g): ref-vs-oracle %.2e, loss %s, sens(last loss) %.2e, sens(last clip logits) %.2e' % (
                WC_LR, errw, [round(float(wc[0]['loss_step%d' % i]), 5) for i in range(3)],
                float(recs[0]['wc/sens/loss_step2']), float(recs[0]['wc/sens/last/out/clip_logits'])))
        np.savez_compressed(os.path.join(GOLD, f'model_{kind}_{net}.npz'), **recs[0])
        print('model', kind, net, 'B', B, 'ref-vs-oracle', err, 'nondist', err2,
              'loss', float(recs[0]['first/total_loss']), float(recs[0][sorted(k for k in recs[0] if k.startswith('loss_step'))[-1]]))


def _loss_worker(rank, world, q, use_ref):
    _init_pg(rank, world, port=29541 + (1 if use_ref else 0))
    ns = harness.load_reference() if use_ref else O
    S = ns.simclr if use_ref else ns
    N, B = 8, 8 // world
    clip = P.procedural_unit_features(N, 2, 128, seed=11)[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    ser = P.procedural_unit_features(N, 2, 2, 64, seed=13)[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    rk = P.procedural_unit_features(N, 2, 2, 64, seed=17)[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    m = S.SimCLR_TimeSeriesV4.__new__(S.SimCLR_TimeSeriesV4)
    torch.nn.Module.__init__(m)
    m.distributed, m.T, m.aligned_T, m.n_series, m.series_dim, m.dim, m.args = world > 1 or use_ref, 0.07, 0.07, 2, 64, 128, ARGS
    m.criterion = torch.nn.CrossEntropyLoss()
    out = {}
    r1 = m.calc_clip_contrast_loss(clip, 2)
    r2 = m.calc_tc_contrast_loss(ser)
    r3 = m.calc_ranking_loss(rk, 2, 'rank_', 0.5)
    (r1['clip_contrast_loss'] + r2['tc_contrast_loss'] + r3['rank_margin_contrast_loss']).backward()
    for r in (r1, r2, r3):
        for k, v in r.items():
            out[k] = v.detach().numpy().copy()
    out['grad_clip'], out['grad_ser'], out['grad_rank'] = clip.grad.numpy().copy(), ser.grad.numpy().copy(), rk.grad.numpy().copy()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def case_losses():
    """Loss-only fixtures from procedural unit features at world_size 1 and 2 (pins GatherLayer
    semantics: per-rank logits, losses and per-rank gradients)."""
    ctx = mp.get_context('spawn')
    rec = {}
    for world in (1, 2):
        res = {}
        for use_ref in (True, False):
            q = ctx.Queue()
            procs = [ctx.Process(target=_loss_worker, args=(r, world, q, use_ref)) for r in range(world)]
            [p.start() for p in procs]
            got = dict(q.get() for _ in range(world))
            [p.join() for p in procs]
            res[use_ref] = got
        for r in range(world):
            err = compare(res[True][r], res[False][r], ('loss', world, r))
            for k, v in res[True][r].items():
                rec[f'w{world}/r{r}/{k}'] = v
            print('losses world', world, 'rank', r, 'ref-vs-oracle', err)
    np.savez_compressed(os.path.join(GOLD, 'losses.npz'), **rec)


def _aug_frames(n_src=12, Hs=20, Ws=26):
    r = np.random.RandomState(4242)
    yy, xx = np.mgrid[0:Hs, 0:Ws]
    base = np.stack([(np.sin(yy / 3.0 + k) * np.cos(xx / 4.0 - k) * 0.5 + 0.5) for k in range(3)], -1)     # smooth structure
    fr = base[None] * r.uniform(0.3, 1.0, size=(n_src, 1, 1, 3)) + r.uniform(-0.15, 0.15, size=(n_src, Hs, Ws, 3))
    return (fr.clip(0, 1) * 255).round().astype(np.uint8)


def _ref_apply(RT, frames, src, win, out_hw, flip, ops_, mean, std):
    """the reference's own functions (utils/transforms.py) on the clip `src`; ops_ = [(code, factors[N])]"""
    from oracle import augment_ref as A
    vid = RT.to_normalized_float_tensor(torch.from_numpy(frames[src]))                  # [N, H, W, C] -> [C, N, H, W]
    i, j, h, w = win
    vid = RT.crop(vid, i, j, h, w)
    if (h, w) != tuple(out_hw):
        vid = RT.resize(vid, tuple(out_hw))
    if flip:
        vid = RT.hflip(vid)
    for code, fac in ops_:
        f = torch.from_numpy(np.asarray(fac, dtype=np.float64))
        if code == A.BRIGHTNESS:
            vid = RT.adjust_brightness(vid, f, 1)
        elif code == A.CONTRAST:
            vid = RT.adjust_contrast(vid, f, 1, 0)
        elif code == A.SATURATION:
            vid = RT.adjust_saturation(vid, f, 1, 0)
        elif code == A.GRAY:
            # random_grayscale :80-88 with its luma taken over the colour axis (as written it indexes the frame axis and
            # asserts unless the clip has exactly 3 frames): gray * mask + vid * (1 - mask)
            m = torch.tensor(np.asarray(fac, dtype=np.float32)).view(1, -1, 1, 1)
            vid = RT.rgb_to_grayscale(vid, 0).unsqueeze(0) * m + vid * (1 - m)
    return RT.normalize(vid, mean, std, channel=0).permute(1, 0, 2, 3).contiguous()      # -> [N, C, H, W]


def case_augment(ref):
    """the augmenting ingest against the reference's tensor-side transforms.  (A) explicit
    parameter rows through every op / order / resize / flip; (B) seeded pipelines: RandomCrop / RandomSizedCrop /
    RandomHorizontalFlip / random_adjust_* consume `random` and `numpy.random` -- the build's parameter classes must
    reproduce the outputs from the same seeds.  (ColorJitter.__call__ itself needs torchvision's Lambda/Compose,
    absent here: its shuffle-then-draw order is restated, not pinned.)"""
    import random
    from oracle import augment_ref as A
    RT = ref.transforms
    frames = _aug_frames()
    H = W = 16
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    rec = {'frames': frames, 'mean': np.float32(mean), 'std': np.float32(std), 'HW': np.int32([H, W])}
    B, C_, S, G = A.BRIGHTNESS, A.CONTRAST, A.SATURATION, A.GRAY
    clips = [                                                       # (src frames, window, flip, [(op, factors)])
        ([0, 1, 2], (2, 5, 16, 16), 0, []),
        ([3, 4, 5], (4, 10, 16, 16), 1, [(B, [0.2, 1.0, 1.8])]),
        ([6, 7, 8], (0, 0, 16, 16), 0, [(C_, [0.2, 1.3, 1.8])]),
        ([9, 10, 11], (3, 3, 16, 16), 1, [(S, [0.0, 0.6, 1.8])]),
        ([0, 4, 8], (1, 2, 16, 16), 0, [(G, [1, 0, 1])]),
        ([1, 5, 9], (0, 0, 20, 26), 0, [(B, [1.5, 0.7, 1.1]), (C_, [0.5, 1.6, 1.0]), (S, [1.7, 0.3, 1.2])]),
        ([2, 6, 10], (2, 1, 13, 22), 1, [(S, [1.7, 0.3, 1.2]), (B, [1.5, 0.7, 1.1]), (C_, [0.5, 1.6, 1.0]), (G, [0, 1, 1])]),
        ([3, 7, 11], (5, 7, 11, 9), 0, [(C_, [1.8, 1.8, 0.1]), (G, [1, 1, 0]), (B, [1.2, 0.4, 1.0])]),
    ]
    rows, want = [], []
    for src, win, flip, ops_ in clips:
        t = np.zeros(len(src), dtype=A.ROW)
        t['src'], t['flip'] = src, flip
        t['crop_i'], t['crop_j'], t['crop_h'], t['crop_w'] = win
        for n in range(len(src)):
            k = 0
            for code, fac in ops_:
                if code == G and not fac[n]:
                    continue
                t['op'][n, k], t['factor'][n, k] = code, fac[n]
                k += 1
        rows.append(t)
        want.append(_ref_apply(RT, frames, src, win, (H, W), flip, ops_, mean, std))
    table = np.concatenate(rows)
    want = torch.stack(want)                                                                   # [clips, T, 3, H, W]
    got = A.augment_ingest(frames, table, len(clips), 3, H, W, mean, std).permute(0, 2, 1, 3, 4)
    err = float((got - want).abs().max())
    print('augment explicit rows: oracle-vs-reference', err)
    assert err < 2e-6, err
    rec['A/table'], rec['A/want'] = table.view(np.uint8).reshape(-1, 64), want.numpy()
    # (B) seeded pipelines
    for tag, sized, consistent in (('crop', False, False), ('sized', True, False), ('sized_consistent', True, True)):
        outs = []
        for seed in (1, 2, 3, 4):
            random.seed(seed)
            np.random.seed(seed)
            src = [(seed + k) % 12 for k in range(4)]
            vid = RT.to_normalized_float_tensor(torch.from_numpy(frames[src]))
            vid = (RT.RandomSizedCrop((H, W)) if sized else RT.RandomCrop((H, W)))(vid)
            vid = RT.RandomHorizontalFlip()(vid)
            vid = RT.random_adjust_saturation(vid, [0.2, 1.8], consistent, 1, 0)
            vid = RT.random_adjust_brightness(vid, [0.2, 1.8], consistent, 1)
            vid = RT.random_adjust_contrast(vid, [0.2, 1.8], consistent, 1, 0)
            outs.append(RT.normalize(vid, mean, std, channel=0).permute(1, 0, 2, 3).float())
        rec['B/%s' % tag] = torch.stack(outs).numpy()                                          # [4 seeds, T=4, 3, H, W]
    # (C) hue: the reference's only hue definition is utils/augmentation.py:adjust_hue_np (uint8 in, uint8 out: float HSV
    # round trip, then truncation to uint8).  The build applies the same arithmetic on floats and does not re-quantise; the
    # fixture keeps the reference's uint8 result, the tests compare floor(255 * x) with it.
    RA = ref.augmentation
    hue_src, hue_fac = [0, 3, 5, 7, 9, 11], [-0.2, -0.07, 0.03, 0.11, 0.2, 0.45]
    t = np.zeros(len(hue_src), dtype=A.ROW)
    t['src'], t['crop_i'], t['crop_j'], t['crop_h'], t['crop_w'] = hue_src, 2, 5, H, W
    t['op'][:, 0], t['factor'][:, 0] = A.HUE, hue_fac
    want_u8 = np.stack([RA.adjust_hue_np(frames[s_][2:2 + H, 5:5 + W], f_) for s_, f_ in zip(hue_src, hue_fac)])   # [6, H, W, 3]
    got = A.augment_ingest(frames, t, len(hue_src), 1, H, W)[:, :, 0].permute(0, 2, 3, 1)                     # [6, H, W, 3] floats
    # a float that lands within rounding of an integer may truncate either way: compare both neighbours
    q = (got * 255.0).numpy()
    diff = np.abs(np.floor(q) - want_u8.astype(np.float64))
    near = np.abs(q - np.round(q)) < 2e-3
    bad = (diff > 0) & ~((diff <= 1) & near)
    print('augment hue: mismatching pixels', int(bad.sum()), 'of', bad.size, '; boundary cases', int(((diff > 0) & near).sum()))
    assert bad.sum() == 0
    rec['C/table'], rec['C/want_u8'] = t.view(np.uint8).reshape(-1, 64), want_u8
    # (D) the SimCLR Gaussian blur (utils/augmentation.py:706-721): the reference hands each finished frame to PIL --
    # transforms.ToPILImage() (= mul(255).byte() of the float frame), ImageFilter.GaussianBlur(radius=sigma), ToTensor() -- with
    # one sigma per clip.  torchvision is absent here, PIL is not: the fixture holds PIL's own uint8 result of those three
    # steps on frames that went through crop / flip / colour ops first; clips with sigma 0 are not blurred (RandomApply).
    from PIL import Image, ImageFilter
    import PIL
    T_ = 3
    dclips = [([0, 1, 2], (2, 5, 16, 16), 0, [], 1.0),
              ([3, 4, 5], (4, 10, 16, 16), 1, [(B, [0.6, 1.0, 1.4])], 0.1),
              ([6, 7, 8], (0, 0, 20, 26), 0, [(C_, [0.5, 1.3, 1.8]), (S, [1.5, 0.6, 1.8])], 2.0),
              ([9, 10, 11], (3, 3, 16, 16), 1, [(S, [0.0, 0.6, 1.8])], 0.0),
              ([1, 5, 9], (2, 1, 13, 22), 0, [(B, [1.5, 0.7, 1.1]), (G, [0, 1, 1])], 0.7371),
              ([2, 6, 10], (5, 7, 11, 9), 1, [], 1.618)]
    rows, sig = [], []
    for src, win, flip, ops_, sigma in dclips:
        t = np.zeros(len(src), dtype=A.ROW)
        t['src'], t['flip'] = src, flip
        t['crop_i'], t['crop_j'], t['crop_h'], t['crop_w'] = win
        for n in range(len(src)):
            k = 0
            for code, fac in ops_:
                if code == G and not fac[n]:
                    continue
                t['op'][n, k], t['factor'][n, k] = code, fac[n]
                k += 1
        rows.append(t)
        sig += [sigma] * len(src)
    dtab = np.concatenate(rows)
    unblurred = A.augment_ingest(frames, dtab, len(dclips), T_, H, W)                          # [N, 3, T, H, W] floats in [0, 1]
    want_u8 = np.zeros((len(dtab), H, W, 3), dtype=np.uint8)
    blur = np.zeros(len(dtab), dtype=A.BLUR)
    for n in range(len(dtab)):
        x = unblurred[n // T_, :, n % T_]                                                       # [3, H, W]
        u8 = x.mul(255).byte().permute(1, 2, 0).numpy()                                         # ToPILImage
        if sig[n] > 0:
            u8 = np.asarray(Image.fromarray(u8).filter(ImageFilter.GaussianBlur(radius=sig[n])))
            blur['radius'][n], blur['ww'][n], blur['fw'][n] = A.box_blur_params(sig[n])
        want_u8[n] = u8
    got = A.augment_ingest(frames, dtab, len(dclips), T_, H, W, blur=blur)                      # the oracle's restatement
    got_u8 = (got * 255).round().to(torch.uint8).permute(0, 2, 3, 4, 1).reshape(-1, H, W, 3).numpy()
    is_blurred = np.float32(sig) > 0
    nbad = int((got_u8 != want_u8)[is_blurred].sum())
    # what the pipeline hands on: blurred frames = PIL's bytes / 255 (ToTensor), the others the float frames as they were
    want_f = unblurred.permute(0, 2, 1, 3, 4).reshape(-1, 3, H, W).clone()
    want_f[torch.from_numpy(is_blurred)] = torch.from_numpy(want_u8[is_blurred]).permute(0, 3, 1, 2).float() / 255
    assert float((got.permute(0, 2, 1, 3, 4).reshape(-1, 3, H, W) - want_f).abs().max()) == 0.0
    print('augment blur: oracle restatement vs PIL %s: %d mismatching bytes of %d' % (PIL.__version__, nbad, want_u8.size))
    assert nbad == 0
    rec['D/table'], rec['D/sigma'], rec['D/T'] = dtab.view(np.uint8).reshape(-1, 64), np.float32(sig), np.int32(T_)
    rec['D/blur'], rec['D/want_u8'], rec['D/want'] = blur.view(np.uint8).reshape(-1, 16), want_u8, want_f.numpy()
    np.savez_compressed(os.path.join(GOLD, 'augment.npz'), **rec)


def case_shapes(ref):
    """The two BASELINE.json clip shapes the other fixtures do not touch: 16-frame clips (configs[1], and the paper's own
    `--seq_len 16`, paper_scripts/paper_table1_k400/pretrain/*.sh) through S3D-G SimCLR_Naked (one full step: outputs,
    gradients), and 32 x 224 x 224 clips (configs[4]) through the 2D3D-ResNet-50 (B = 1: pooled features and a strided
    sample of the feature map)."""
    _init_pg()
    rec = {}
    # --- 16 x 112 x 112
    B = 2
    block = P.procedural_clips(B, 2, T=16, H=112, W=112)
    recs = []
    for ns in (ref, O):
        torch.manual_seed(0)
        m = build(ns, 'simclr_naked', 's3dg', True)
        P.procedural_init(m)
        recs.append(run_model(m, block, steps=1, np_seed=1234))
    err = compare(recs[0], recs[1], ('t16',))
    torch.manual_seed(0)
    m = build(ref, 'simclr_naked', 's3dg', True)
    P.procedural_init(m)
    noise = torch.from_numpy(np.random.RandomState(99).standard_normal(block.numel())).float().reshape(block.shape)
    pert = run_model(m, block * (1 + 1e-6 * noise), steps=1, np_seed=1234)
    for k, v in gsample_noise_floor(ref, 'simclr_naked', 's3dg', True, block).items():
        rec['t16/f64/first/gsample/' + k] = np.array(v)
    for k, v in recs[0].items():
        rec['t16/' + k] = v
        if k in pert and 'labels' not in k:
            a, b = np.asarray(v, dtype=np.float64), np.asarray(pert[k], dtype=np.float64)
            rec['t16/sens/' + k] = np.array(float(np.max(np.abs(a - b))))
    print('shapes: s3dg simclr_naked 16x112x112 B', B, 'ref-vs-oracle', err, 'loss', float(recs[0]['first/total_loss']))
    # --- 32 x 224 x 224 through r50
    x = P.procedural_clips(1, 1, T=32, H=224, W=224)[:, 0]
    outs = []
    for sel in (ref.select_backbone, O.select_backbone):
        m, _ = sel('r50')
        P.procedural_init(m).train()
        with torch.no_grad():
            outs.append(m(x))
    err = float((outs[0] - outs[1]).abs().max())
    assert err < 1e-5, err
    with torch.no_grad():
        m2, _ = ref.select_backbone('r50')
        P.procedural_init(m2).train()
        y64 = m2.double()(x.double())
    rec['r50_224/shape'] = np.array(outs[0].shape)
    rec['r50_224/pooled'] = outs[0].mean(dim=(2, 3, 4)).numpy()
    rec['r50_224/feat_sample'] = outs[0].reshape(-1)[::997].numpy().copy()
    rec['r50_224/fp32_vs_fp64'] = np.array(float((outs[0].double() - y64).abs().max() / y64.abs().max()))
    print('shapes: r50 32x224x224', tuple(outs[0].shape), 'ref-vs-oracle', err, 'fp32-vs-fp64', float(rec['r50_224/fp32_vs_fp64']))
    np.savez_compressed(os.path.join(GOLD, 'shapes.npz'), **rec)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['backbones', 'models', 'losses', 'eval', 'clf', 'extra', 'augment', 'shapes']
    if 'losses' in which:
        case_losses()
    ref = harness.load_reference()
    if 'backbones' in which:
        case_backbones(ref)
    if 'eval' in which:
        case_eval(ref)
    if 'clf' in which:
        case_classifier_train(ref)
    if 'extra' in which:
        case_backbones_extra(ref)
    if 'augment' in which:
        case_augment(ref)
    if 'shapes' in which:
        case_shapes(ref)
    if 'models' in which:
        case_models(ref)
    if 'models_f64_losses' in which:
        case_models_add_f64_step_losses(ref)


if __name__ == '__main__':
    main()
