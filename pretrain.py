#!/usr/bin/env python
"""pretrain.py -- DualVar self-supervised pretraining on MI355X (drop-in for the reference entry script).

Same command line as the reference (pretrain.py:90-164, incl. the `--series_mode` spelling its launch scripts
use and the flags its code reads but never defines: SURVEY.md D4/D5), same model factory (`get_model`), the
same train-loop body (pretrain.py:394-466): forward -> sum of every '*loss' head -> backward -> SGD step, with
the per-head loss / top-1 meters.  One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).

What differs, deliberately:
  * data: `--dataset synthetic` (the metric's workload; the JPEG/LMDB loaders of dataset/local_dataset.py and the
    PIL augmentations are the CPU data pipeline, out of scope -- SURVEY.md 2.1 rows 13-14); `--dataset
    synthetic-frames` feeds decoded uint8 128x171 frames and augments them ON THE GPU inside the ingest kernel
    (crop / flip / colour jitter / grayscale of utils/transforms.py, dualvar_amd.utils.transforms.FrameBatch);
  * Normalize (utils/transforms.py) is fused into the ingest kernel instead of a separate GPU pass;
  * SyncBatchNorm / DDP are the engine's own collectives (dualvar_amd/parallel.py), not module wrappers;
  * accuracy meters read the positive's rank emitted by the loss kernels (no topk launch, no extra sync):
    all scalars of a step come back in ONE device->host copy.
"""
import os as _os
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')   # kernel arguments in device memory: -4 % step time (read when HIP loads)
import argparse
import os
import random
import sys
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dualvar_amd.model import MoCo_Naked, MoCo_TimeSeriesV4, SimCLR_Naked, SimCLR_TimeSeriesV4  # noqa: E402
from dualvar_amd.utils.utils import AverageMeter, ProgressMeter, neq_load_customized, save_checkpoint  # noqa: E402

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def get_model(args):
    """pretrain.py:61-77"""
    kw = dict(n_series=args.n_series, series_dim=args.series_dim, series_T=args.series_T, aligned_T=args.aligned_T,
              mode=args.mode, args=args)
    if args.model == 'moco_naked':
        return MoCo_Naked(args.net, args.moco_dim, args.moco_k, args.moco_m, args.moco_t, args.distributed)
    if args.model == 'moco_timeseriesv4':
        return MoCo_TimeSeriesV4(args.net, args.moco_dim, args.moco_k, args.moco_m, args.moco_t, args.distributed, **kw)
    if args.model == 'simclr_naked':
        return SimCLR_Naked(args.net, args.moco_dim, args.moco_t, args.distributed)
    if args.model == 'simclr_timeseriesv4':
        return SimCLR_TimeSeriesV4(args.net, args.moco_dim, args.moco_t, args.distributed, **kw)
    raise NotImplementedError(args.model)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    # model
    p.add_argument('--net', default='r21d', type=str)
    p.add_argument('--model', default='simclr_timeseriesv4', type=str)
    p.add_argument('--series_dim', default=64, type=int)
    p.add_argument('--n_series', default=2, type=int)
    p.add_argument('--shufflerank_theta', default=0.05, type=float)
    p.add_argument('--series_T', default=0.07, type=float)
    p.add_argument('--aligned_T', default=0.07, type=float)
    p.add_argument('--mode', '--series_mode', dest='mode', default='clip-sr-tc', type=str, choices=['clip-sr-tc', 'clip-sr'])
    p.add_argument('--moco-dim', default=128, type=int)
    p.add_argument('--moco-k', default=2048, type=int)
    p.add_argument('--moco-m', default=0.999, type=float)
    p.add_argument('--moco-t', default=0.07, type=float)
    # dataset
    p.add_argument('--dataset', default='synthetic', type=str)
    p.add_argument('--seq_len', default=16, type=int)
    p.add_argument('--num_seq', default=2, type=int)
    p.add_argument('--ds', default=4, type=int)
    p.add_argument('--img_dim', default=112, type=int)
    p.add_argument('--gpu', default=None, type=int)
    p.add_argument('-j', '--workers', default=16, type=int)
    p.add_argument('--seed', default=0, type=int)
    p.add_argument('--aug_temp_consist', action='store_true')
    p.add_argument('--aug_series', action='store_true')
    p.add_argument('--rand_flip', action='store_true')
    # optimizer
    p.add_argument('--optim', default='sgd', type=str)
    p.add_argument('--batch_size', default=32, type=int)
    p.add_argument('--lr', default=0.03, type=float)
    p.add_argument('--wd', default=5e-4, type=float)
    p.add_argument('--epochs', default=200, type=int)
    p.add_argument('--start_epoch', default=0, type=int)
    p.add_argument('--schedule', default=[120, 160], nargs='*', type=int)
    # log
    p.add_argument('--print_freq', default=20, type=int)
    p.add_argument('--eval_freq', default=5, type=int)
    p.add_argument('--save_freq', default=5, type=int)
    p.add_argument('--resume', default='', type=str)
    p.add_argument('--pretrain', default='', type=str)
    p.add_argument('--prefix', default='pretrain', type=str)
    p.add_argument('--name_prefix', default='', type=str)
    # parallel
    p.add_argument('--world-size', default=-1, type=int)
    p.add_argument('--rank', default=-1, type=int)
    p.add_argument('--dist-url', default='env://', type=str)
    p.add_argument('--dist-backend', default='nccl', type=str)
    p.add_argument('--multiprocessing-distributed', action='store_true')
    p.add_argument('--local_rank', '--local-rank', dest='local_rank', default=-1, type=int)
    # flags the reference reads without defining
    p.add_argument('--n_proto', default=1, type=int)
    p.add_argument('--n_block', default=1, type=int)
    p.add_argument('--aug_temp_grad_consist', action='store_true')
    p.add_argument('--visualize', action='store_true')
    p.add_argument('--test', default='', type=str)
    # this build
    p.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                   help='HIP engine arithmetic: fp32 = the reference\'s (parity mode); bf16 = bf16 activation storage (throughput mode)')
    p.add_argument('--steps', default=0, type=int, help='stop every epoch after this many iterations (0 = whole epoch)')
    p.add_argument('--epoch_size', default=1024, type=int, help='synthetic samples per epoch (whole job)')
    p.add_argument('--warm_steps', default=20, type=int, help='steps excluded from the steady-state clips/s of an epoch')
    return p.parse_args(argv)


class SyntheticClips(torch.utils.data.Dataset):
    """batch['seq'] as the reference datasets yield it: [3, num_seq*n_proto*seq_len, H, W] float in [0,1]."""

    def __init__(self, args, length):
        self.shape = (3, args.num_seq * args.n_proto * args.seq_len, args.img_dim, args.img_dim)
        self.length, self.seed = length, args.seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return {'seq': torch.rand(self.shape, generator=g)}


class SyntheticFrames(torch.utils.data.Dataset):
    """batch['frames']: what a video decoder hands over -- uint8 [seq_len, 128, 171, 3] (A.Scale((128, 171)) of
    pretrain.py:494); every view of the sample is cut from these frames by the augmenting ingest.
    The decoder itself is out of scope: samples are drawn from a pool of pre-generated "videos", so a
    worker's cost per sample is what the REAL pipeline also pays after decoding -- the augmentation draws.  With `transform`
    the DataLoader WORKER draws the sample's augmentation (the reference augments in its workers too, pretrain.py:491-564) and
    returns it as table rows (`aug`, `blur`: dualvar_amd.utils.transforms.AUG_ROW / AUG_BLUR), `views` views per sample; the pixels
    are only touched on the GPU."""

    def __init__(self, args, length, transform=None, views=1, pool=64):
        self.shape, self.length, self.seed = (args.seq_len, 128, 171, 3), length, args.seed
        rs = np.random.RandomState(args.seed)
        self.pool = torch.from_numpy(rs.randint(0, 256, (min(length, pool),) + self.shape, dtype=np.uint8))
        self.transform, self.views, self.size = transform, views, (args.img_dim, args.img_dim)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        out = {'frames': self.pool[i % self.pool.shape[0]]}
        if self.transform is not None:
            from dualvar_amd.utils.transforms import ClipState
            L, Hs, Ws = self.shape[:3]
            rows, blurs = [], []
            for _ in range(self.views):
                st = self.transform(ClipState(range(L), Hs, Ws))
                rows.append(st.rows(*self.size))
                blurs.append(st.blur_rows())
            out['aug'] = torch.from_numpy(np.concatenate(rows).view(np.uint8).copy())
            out['blur'] = torch.from_numpy(np.concatenate(blurs).view(np.uint8).copy())
        return out


def collate_frames(samples):
    """default collate + the source-frame indices of sample b's augmentation rows moved to its place in the batch (b * L)"""
    batch = torch.utils.data.default_collate(samples)
    if 'aug' in batch:
        B, L = batch['frames'].shape[:2]
        rows = batch['aug'].numpy().view(np.int32).reshape(B, -1, 16)            # AUG_ROW: 16 words, word 0 = source frame
        rows[:, :, 0] += (np.arange(B, dtype=np.int32) * L)[:, None]
        blur = batch['blur'].numpy().view(np.uint32).reshape(B, -1, 4)
        batch['has_blur'] = bool(blur[:, :, 1].any())                             # AUG_BLUR.ww: 0 = this frame is not blurred
    return batch


def seed_worker(worker_id):
    """the augmentation classes draw from `random` / `numpy.random` (as the reference's do, utils/augmentation.py): one stream per
    DataLoader worker, derived from torch's per-worker seed (utils/utils.py:worker_init_fn of the reference)"""
    s_ = torch.initial_seed() % (1 << 31)
    random.seed(s_)
    np.random.seed(s_)


class DevicePrefetcher:
    """Hands out batches that are already on the GPU: batch i + 1 is copied host -> device from pinned memory on a side HIP stream
    while step i computes (the reference copies in line, `.cuda(non_blocking=True)`, pretrain.py:396-400)."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device('cuda', device)
        self.stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)

        def load():
            try:
                b = next(it)
            except StopIteration:
                return None
            with torch.cuda.stream(self.stream):
                return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}
        nxt = load()
        while nxt is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            for v in nxt.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
            batch, nxt = nxt, load()
            yield batch


def gpu_transform(args):
    """the base transform of pretrain.py:500-509 in its tensor-side form (utils/transforms.py): random crop, optional
    flip, colour jitter 0.8 / 0.8 / 0.8 / hue 0.2 with p = 0.8 (temporally consistent under --aug_temp_consist; hue by
    utils/augmentation.py:adjust_hue_np's arithmetic); then, with p = 0.5, the SimCLR Gaussian blur (sigma in [0.1, 2], one
    per clip) exactly as PIL computes it (utils/augmentation.py:706-721)"""
    from dualvar_amd.utils import transforms as T
    steps = [T.RandomCrop((args.img_dim, args.img_dim))]
    if args.rand_flip:
        steps.append(T.RandomHorizontalFlip())
    steps.append(T.ColorJitter(0.8, 0.8, 0.8, consistent=args.aug_temp_consist, p=0.8 * 0.8, hue=0.2))
    steps.append(T.RandomApply([T.GaussianBlur([.1, 2.], seq_len=args.seq_len)], p=0.5))
    return T.Compose(steps)


def main(args):
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    random.seed(args.seed)
    if 'WORLD_SIZE' in os.environ and args.world_size < 0:
        args.world_size = int(os.environ['WORLD_SIZE'])
    args.distributed = args.world_size > 1 or args.multiprocessing_distributed
    ngpus = torch.cuda.device_count()
    if args.multiprocessing_distributed:
        args.world_size = ngpus * max(args.world_size, 1)
        torch.multiprocessing.spawn(main_worker, nprocs=ngpus, args=(ngpus, args))
    else:
        main_worker(args.gpu, ngpus, args)


def main_worker(gpu, ngpus_per_node, args):
    if args.distributed:
        if args.local_rank != -1:
            args.rank, args.gpu = args.local_rank, args.local_rank
        elif 'SLURM_PROCID' in os.environ:
            args.rank = int(os.environ['SLURM_PROCID'])
            args.gpu = args.rank % max(torch.cuda.device_count(), 1)
        elif 'RANK' in os.environ:
            args.rank = int(os.environ['RANK'])
            args.gpu = int(os.environ.get('LOCAL_RANK', args.rank % max(ngpus_per_node, 1)))
        elif args.multiprocessing_distributed:
            args.rank, args.gpu = gpu, gpu
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        torch.cuda.set_device(args.gpu)
        dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                                rank=args.rank, device_id=torch.device('cuda', args.gpu))
    else:
        args.rank, args.gpu = 0, (args.gpu if args.gpu is not None else 0)
        torch.cuda.set_device(args.gpu)
    args.print = is_printing_rank(args.distributed, args.rank)
    args.img_path, args.model_path, args.exp_path = set_path(args)
    args.logger = DistLogger(os.path.join(args.exp_path, 'log.txt'), args.print)
    args.iteration = 1

    from dualvar_amd.optim import SGD
    from dualvar_amd.parallel import GradSync
    model = get_model(args)
    model.set_compute_dtype(args.dtype)
    model.set_input_normalization(MEAN, STD)            # T.Normalize(channel=1) of pretrain.py:280-282, fused
    model.cuda(args.gpu)
    params = [{'params': [p]} for p in model.parameters() if p.requires_grad]     # pretrain.py:262-271
    gsync = GradSync() if args.distributed else None
    if gsync is not None:
        gsync.attach(model)         # single-pass objectives: bucket-wise all-reduce from inside the backward pass
    optimizer = SGD(params, lr=args.lr, weight_decay=args.wd, momentum=0.9, stores=model.stores(), grad_sync=gsync)

    per_rank = max(args.epoch_size // max(args.world_size, 1), args.batch_size)
    n_samples = per_rank * max(args.world_size, 1)
    if args.dataset == 'synthetic-frames':
        # every view's augmentation is drawn in the DataLoader workers and applied on the GPU by the ingest kernel
        args.gpu_transform = gpu_transform(args)
        dataset = SyntheticFrames(args, n_samples, transform=args.gpu_transform, views=args.num_seq * args.n_proto)
    else:
        dataset = SyntheticClips(args, n_samples)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=True) if args.distributed else None
    nw = min(args.workers, 4)
    loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                                         num_workers=nw, pin_memory=True, drop_last=True, worker_init_fn=seed_worker,
                                         collate_fn=collate_frames if args.dataset == 'synthetic-frames' else None,
                                         persistent_workers=nw > 0, prefetch_factor=4 if nw > 0 else None)
    loader = DevicePrefetcher(loader, args.gpu)

    best_acc = 0
    if args.resume and os.path.isfile(args.resume):
        ck = torch.load(args.resume, map_location='cpu', weights_only=True)
        args.start_epoch, args.iteration, best_acc = resume_position(ck)
        try:
            model.load_state_dict(ck['state_dict'])
        except Exception:
            neq_load_customized(model, ck['state_dict'], verbose=True, args=args)
        if 'optimizer' in ck:
            try:
                n = optimizer.load_state_dict(ck['optimizer'])
                args.logger.info('optimizer state restored (%d momentum buffers)' % n)
            except Exception as e:
                args.logger.info('optimizer state not restored: %s' % e)
    elif args.pretrain and os.path.isfile(args.pretrain):
        ck = torch.load(args.pretrain, map_location='cpu', weights_only=True)
        neq_load_customized(model, ck['state_dict'], verbose=True, args=args)

    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=args.schedule, gamma=0.1,
                                                     last_epoch=args.start_epoch - 1)
    for epoch in range(args.start_epoch, args.epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        np.random.seed(epoch)
        random.seed(epoch)
        _, train_acc = train_one_epoch(loader, model, optimizer, scheduler, None, epoch, args)
        if (epoch % args.save_freq == 0 or epoch == args.epochs - 1) and args.print:
            is_best = train_acc > best_acc
            best_acc = max(train_acc, best_acc)
            save_checkpoint({'epoch': epoch, 'state_dict': model.state_dict(), 'best_acc': best_acc,
                             'optimizer': optimizer.state_dict(), 'iteration': args.iteration}, is_best,
                            gap=args.save_freq, filename=os.path.join(args.model_path, 'epoch%d.pth.tar' % epoch),
                            keep_all=True)
    args.logger.info('Training from ep %d to ep %d finished' % (args.start_epoch, args.epochs))
    if args.distributed:
        torch.cuda.synchronize()
        from dualvar_amd import rccl
        rccl.destroy_all()
        dist.destroy_process_group()


def is_printing_rank(distributed, rank):
    """pretrain.py:225 (`args.print = args.gpu == 0 or not args.distributed`): the process that logs and saves
    checkpoints -- rank 0 of a distributed job; a single-process run always, whatever --gpu it was given."""
    return (rank == 0) if distributed else True


def resume_position(ck):
    """pretrain.py:290-292,343: a checkpoint's 'epoch' is the LAST FINISHED epoch -> (start_epoch, iteration, best_acc)"""
    return int(ck['epoch']) + 1, ck.get('iteration', 1), float(ck.get('best_acc', 0))


def train_one_epoch(data_loader, model, optimizer, scheduler, transforms_cuda, epoch, args,
                    model_head=None, head_optimizer=None, head_scheduler=None):
    """pretrain.py:364-489 (signature as the reference defines it)."""
    batch_time, data_time = AverageMeter('Time', ':.2f'), AverageMeter('Data', ':.2f')
    issue_time = AverageMeter('Issue', ':.4f')     # host time to build the input and queue the step's launches (no GPU wait inside)
    losses_meters, acc_meters = OrderedDict(clip=AverageMeter('VLoss', ':.4f')), OrderedDict(clip=AverageMeter('Vacc@1', ':.4f'))
    progress = ProgressMeter(len(data_loader), [batch_time, data_time], prefix='Epoch:[{}/{}] lr:{} '.format(
        epoch, args.epochs, optimizer.param_groups[0]['lr']), logger=args.logger)
    model.train()
    nv = args.num_seq * args.n_proto

    def tr(x):          # pretrain.py:386-389; Normalize runs inside the ingest kernel, only the view change is left
        B = x.size(0)
        return x.view(B, 3, nv, args.seq_len, args.img_dim, args.img_dim).transpose(1, 2).contiguous()

    tic = end = time.time()
    clips = 0
    warm = min(args.warm_steps, max(len(data_loader) // 4, 0)) if not args.steps else min(args.warm_steps, args.steps // 4)
    t_steady, clips_steady = None, 0
    pending = None          # (device scalars, heads, B) of the previous step: read back AFTER this step's launches are queued

    # The scalars of a step go device -> host on a COPY STREAM, behind an event recorded where they are produced: a `.cpu()` on
    # the main stream would queue behind every launch of the NEXT step (already issued when the copy is requested) and the
    # host would find the GPU idle each time it comes back from it (measured: 23.1 instead of 20.x ms per step).
    copy_stream = torch.cuda.Stream(args.gpu)
    host_bufs = [None, None]

    def stage(dev_vec, heads, B, slot):
        if host_bufs[slot] is None or host_bufs[slot].numel() != dev_vec.numel():
            host_bufs[slot] = torch.empty(dev_vec.numel(), dtype=torch.float32).pin_memory()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(args.gpu))
        dev_vec.record_stream(copy_stream)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready)
            host_bufs[slot].copy_(dev_vec, non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy_stream)
        return (done, host_bufs[slot], heads, B)

    def drain(p):
        done, buf, heads, B = p
        done.synchronize()
        host = buf.tolist()
        for i, h in enumerate(heads):
            if h not in losses_meters:
                losses_meters[h], acc_meters[h] = AverageMeter(f'{h}_loss', ':.3f'), AverageMeter(f'{h}_acc', ':.3f')
            losses_meters[h].update(host[i], B)
            acc_meters[h].update(host[len(heads) + i], B)

    for idx, batch in enumerate(data_loader):
        data_time.update(time.time() - end)
        if 'frames' in batch:        # decoded frames: every view is augmented inside the ingest kernel
            from dualvar_amd.utils.transforms import FrameBatch
            fr = batch['frames']                                                         # [B, L, Hs, Ws, 3] uint8, on the GPU
            if not fr.is_cuda:
                fr = fr.cuda(args.gpu, non_blocking=True)
            if 'aug' in batch:       # rows drawn by the DataLoader workers (SyntheticFrames with a transform)
                shape = (fr.size(0), nv, 3, args.seq_len, args.img_dim, args.img_dim) if nv > 1 else \
                        (fr.size(0), 3, args.seq_len, args.img_dim, args.img_dim)
                input_seq = FrameBatch(fr.view(-1, *fr.shape[2:]), batch['aug'].cuda(args.gpu).view(-1), shape,
                                       blur=batch['blur'].cuda(args.gpu).view(-1) if batch.get('has_blur', True) else None)
            else:
                L_ = fr.size(1)
                input_seq = FrameBatch.build(fr.view(-1, *fr.shape[2:]), [list(range(b * L_, b * L_ + args.seq_len)) for b in range(fr.size(0))],
                                             args.gpu_transform, (args.img_dim, args.img_dim), views=nv)
        else:
            x = batch['seq']
            input_seq = tr(x if x.is_cuda else x.cuda(args.gpu, non_blocking=True))
        B = input_seq.size(0)
        ret = model(input_seq)
        loss = 0
        heads = []
        if 'clip_contrast_loss' in ret:
            loss = ret['clip_contrast_loss']
            heads.append('clip')
        for key in ret:
            if 'loss' in key and 'clip' not in key:
                loss = loss + ret[key]
                heads.append(key.replace('_contrast_loss', '').replace('_loss', ''))
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        clips += B * input_seq.size(1)

        # one device->host copy for every scalar of the step (the reference does 2 .item() per head, pretrain.py:410-411,431-432),
        # taken one step LATE: the copy of step i waits for the GPU, so it is issued after step i + 1 has been queued -- the
        # host never runs dry of work it could be launching
        scal = [ret[(h + '_contrast_loss') if (h + '_contrast_loss') in ret else (h + '_loss')].detach().reshape(1) for h in heads]
        top1 = []
        for h in heads:
            if (h + '_rank0') in ret:
                top1.append((ret[h + '_rank0'] < 1).float().mean().reshape(1))
            elif (h + '_logits') in ret:
                lg = ret[h + '_logits']
                top1.append((lg[:, 1:].max(dim=1).values < lg[:, 0]).float().mean().reshape(1))
            else:
                top1.append(torch.zeros(1, device=loss.device))
        new_pending = stage(torch.cat(scal + top1), heads, B, idx & 1)
        issue_time.update(time.time() - end)
        if pending is not None:
            drain(pending)
        pending = new_pending
        batch_time.update(time.time() - end)
        end = time.time()
        if idx + 1 == warm:
            torch.cuda.synchronize()
            t_steady, clips_steady = time.time(), clips
        if (idx + 1) % args.print_freq == 0 and args.print and losses_meters['clip'].count > 0:     # (the meters run one step late)
            progress.meters = [batch_time, data_time, issue_time] + list(losses_meters.values()) + list(acc_meters.values())
            progress.display(idx)
        args.iteration += 1
        if args.steps and idx + 1 >= args.steps:
            break
    if pending is not None:
        drain(pending)
    torch.cuda.synchronize()
    dt = time.time() - tic
    world = max(args.world_size, 1)
    args.logger.info('Epoch: [{0}/{1}]\tT-epoch:{t:.2f}\tLoss:{loss:.4f}\tclips/s (whole job):{cps:.1f}'.format(
        epoch, args.epochs, t=dt, loss=sum(m.avg for m in losses_meters.values()), cps=clips * world / max(dt, 1e-9)))
    if t_steady is not None and clips > clips_steady:
        # steady state: after the first `warm` steps (plan building, first-touch allocation, DataLoader start-up)
        args.steady_clips_per_s = (clips - clips_steady) * world / max(time.time() - t_steady, 1e-9)
        args.logger.info('Epoch: [{0}/{1}]\tsteady-state clips/s (whole job, steps {w}..{n}):{cps:.1f}\tdata wait per step:{dw:.2f} ms'.format(
            epoch, args.epochs, w=warm, n=idx + 1, cps=args.steady_clips_per_s, dw=data_time.avg * 1e3) +
                         '\thost issue per step:%.2f ms' % (issue_time.avg * 1e3))
    scheduler.step()
    args.lr = optimizer.param_groups[0]['lr']
    return losses_meters['clip'].avg, acc_meters['clip'].avg


class DistLogger:
    def __init__(self, log_file, do_print=True):
        self.print = do_print
        self.fh = open(log_file, 'a') if do_print else None

    def info(self, content):
        if self.print:
            line = time.strftime('%Y-%m-%d %H:%M:%S ') + str(content)
            print(line, flush=True)
            self.fh.write(line + '\n')
            self.fh.flush()


def set_path(args):
    """pretrain.py:567-591 (directory naming kept)"""
    if args.resume:
        exp_path = os.path.dirname(os.path.dirname(args.resume))
    else:
        name = '{a.name_prefix}{a.model}_k{a.moco_k}_{a.dataset}-{a.img_dim}_{a.net}_bs{a.batch_size}_lr{a.lr}_seq{a.num_seq}_len{a.seq_len}_ds{a.ds}'.format(a=args)
        exp_path = os.path.join('log-' + args.prefix, name)
    img_path, model_path = os.path.join(exp_path, 'img'), os.path.join(exp_path, 'model')
    if args.rank in (0, -1):
        os.makedirs(img_path, exist_ok=True)
        os.makedirs(model_path, exist_ok=True)
    else:
        time.sleep(0.2)
        os.makedirs(exp_path, exist_ok=True)
    return img_path, model_path, exp_path


if __name__ == '__main__':
    main(parse_args())
