#!/usr/bin/env python
"""bench.py -- clips/sec of the DualVar pretrain step (S3D-G 8x112x112 SimCLR) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          (no launcher: starts the N ranks itself, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = ingest + forward + NT-Xent + backward + (gradient all-reduce) + SGD on one synthetic batch per GPU
(weak scaling: per-GPU batch fixed).  Prints ONE JSON line on rank 0 (contract in the task statement), including
  value        : the step in fp32 -- the reference's arithmetic and the mode with 1e-3 loss parity; `secondary` carries the
                 same step with bf16 storage (its own roofline), clearly labelled
  roofline     : the kernel with the largest share of the step, its algorithmic bytes (or flops) per launch over
                 its HIP-event duration measured inside the timed region
  cpu_baseline : the CPU oracle (oracle/torch_ref.py, validated == the reference) timed on this host on a bounded
                 sample of the same workload.
"""
import os as _os
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')   # kernel arguments in device memory: -4 % step time (read when HIP loads)
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from dualvar_amd import _lib as _dv_lib  # noqa: E402   (pure Python at import: no GPU call)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# dense MFMA peaks per dtype (MI355X_MICROARCH.md).  fp32 is priced against the guide's fp32 matrix peak, 157.3 TFLOP/s.  The
# step's fp32 products actually run as six bf16 partial products on the bf16 matrix cores (csrc/conv.hip: split3), whose
# roof for ALGORITHMIC fp32 flops is 2500 / 6 = 416.7 TFLOP/s: reported next to it as roofline.split_peak / frac_of_split_peak.
MFMA_PEAK_TF = {'bf16': 2500.0, 'fp32': 157.3, 'fp8pw': 2500.0}
F32_SPLIT_PEAK_TF = 2500.0 / 6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    # fp32 is the reference's arithmetic (no autocast / half anywhere in lzhangbj/DualVar) and the mode whose outputs match
    # the reference within the north-star 1e-3 (tests/test_models_gpu.py): it is the headline `value`.  bf16 storage is
    # reported next to it as `secondary` (--secondary none switches that leg off).
    ap.add_argument('--dtype', default='fp32', choices=['bf16', 'fp32', 'fp8pw'],
                    help="fp8pw: bf16 storage with the bottleneck 1x1x1 convs on the fp8 matrix cores (BASELINE configs[4], --net r50)")
    ap.add_argument('--secondary', default='auto', choices=['auto', 'bf16', 'fp32', 'fp8pw', 'none'],
                    help="second timed leg in the other storage dtype (auto: bf16 when --dtype is fp32)")
    ap.add_argument('--net', default='s3dg')
    ap.add_argument('--model', default='simclr_naked',
                    choices=['simclr_naked', 'simclr_timeseriesv4', 'moco_naked', 'moco_timeseriesv4'])
    ap.add_argument('--batch', type=int, default=64, help='samples per GPU (each sample = 2 or 3 clip views)')
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--size', type=int, default=112)
    ap.add_argument('--moco-k', type=int, default=65536)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dump-launches', default='', help='write a per-launch timing table (calibration step) to this file')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--cpu-steps', type=int, default=2)
    ap.add_argument('--launch-check', action='store_true',
                    help='only exercise the N-rank launch / rendezvous / watchdog plumbing (no GPU work): every rank joins the process '
                         'group, one all-reduce, rank 0 prints {"launch_check": true, "n_gpus": N}')
    ap.add_argument('--hang-check', type=float, default=0.0,
                    help='with --launch-check: rank 1 never joins the all-reduce, and the watchdog deadline is this many seconds '
                         '(tests the deadline path)')
    return ap.parse_args()


def build_model(args, distributed):
    import types
    from dualvar_amd import model as M
    a = types.SimpleNamespace(shufflerank_theta=0.05)
    if args.model == 'simclr_naked':
        return M.SimCLR_Naked(args.net, 128, 0.07, distributed)
    if args.model == 'simclr_timeseriesv4':
        return M.SimCLR_TimeSeriesV4(args.net, 128, 0.07, distributed, args=a)
    if args.model == 'moco_naked':
        return M.MoCo_Naked(args.net, 128, args.moco_k, 0.999, 0.07, distributed)
    return M.MoCo_TimeSeriesV4(args.net, 128, args.moco_k, 0.999, 0.07, distributed, args=a)


def total_loss(ret):
    loss = ret['clip_contrast_loss']
    for k in ret:
        if 'loss' in k and 'clip' not in k:
            loss = loss + ret[k]
    return loss


def kernel_of(label):
    """the GPU function (template instantiation) behind a launch label: the LDS-staged kernel's forward and data gradient are ONE
    instantiation (the tap displacement is a runtime sign), and so is a data gradient with or without the fused BatchNorm-backward
    reduce (a runtime pointer) -- what rocprofv3 lists as one kernel is one entry here.  `+bn_in` / `+bn_bwd_apply` stay: those
    are template parameters."""
    label = label.replace('+bn_reduce', '')
    if label.startswith('conv_tap<'):
        label = label.replace('FWD,', '').replace('DGRAD,', '')
    return label


class KernelTimer:
    """HIP-event timing of plan launches on the stream they are launched on (torch's current stream)."""

    def __init__(self, only=None):
        self.only = only               # None: time every launch (calibration); else a kernel-name
        self.rec = []                  # (launch, start, end)

    def __call__(self, launch, stream, stream_obj=None):
        """stream: raw hipStream_t the launch goes to; stream_obj: its torch stream when it is not the current one
        (the plan issues weight gradients on a side stream) -- the events are recorded on the launch's own stream"""
        if self.only is not None and kernel_of(launch.kname) != self.only:
            launch(stream)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if stream_obj is None:
            a.record()
            launch(stream)
            b.record()
        else:
            a.record(stream_obj)
            launch(stream)
            b.record(stream_obj)
        self.rec.append((launch, a, b))

    def summary(self, key=lambda k: k):
        """{key(kname): [n, total_ms, total_bytes, total_flops]} (call after a device sync)"""
        out = {}
        for l, a, b in self.rec:
            e = out.setdefault(key(l.kname), [0, 0.0, 0, 0])
            e[0] += 1
            e[1] += a.elapsed_time(b)
            e[2] += l.bytes
            e[3] += l.flops
        return out


def all_plans(model):
    for m in model.modules():
        if hasattr(m, '_plans'):
            for lst in m._plans.values():
                for p in lst:
                    yield p


def cpu_baseline(args, V):
    """The oracle restatement (== reference, oracle/gen_golden.py) on this host's cores: same model, same clip
    shape, bounded batch / steps."""
    from oracle import torch_ref as O
    import types
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, int(os.environ.get('DUALVAR_CPU_THREADS', 16))))   # a 1-GPU box has a 16-core share
    torch.set_num_threads(ncpu)
    print(f'[bench] cpu baseline: {ncpu} threads', file=sys.stderr, flush=True)
    a = types.SimpleNamespace(shufflerank_theta=0.05)
    torch.manual_seed(0)
    if args.model == 'simclr_naked':
        m = O.SimCLR_Naked(args.net, 128, 0.07, False)
    elif args.model == 'simclr_timeseriesv4':
        m = O.SimCLR_TimeSeriesV4(args.net, 128, 0.07, False, args=a)
    elif args.model == 'moco_naked':
        m = O.MoCo_Naked(args.net, 128, min(args.moco_k, 4096), 0.999, 0.07, False)
    else:
        m = O.MoCo_TimeSeriesV4(args.net, 128, min(args.moco_k, 4096), 0.999, 0.07, False, args=a)
    m.train()
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.003, momentum=0.9, weight_decay=1e-4)
    B = args.cpu_batch
    block = torch.randn(B, V, 3, args.frames, args.size, args.size, generator=torch.Generator().manual_seed(1234))
    best = None
    for it in range(args.cpu_steps + 1):
        t0 = time.time()
        O.train_step(m, block, opt)
        dt = time.time() - t0
        print(f'[bench] cpu step {it}: {dt:.2f} s', file=sys.stderr, flush=True)
        if it > 0:
            best = dt if best is None else min(best, dt)
    return {'value': round(B * V / best, 2), 'unit': 'clips/s', 'cores': ncpu, 'kind': 'port',
            'sample': f'oracle/torch_ref.py {args.model}/{args.net} fp32, batch {B}x{V} clips of {args.frames}x{args.size}x{args.size}, '
                      f'best of {args.cpu_steps} full train steps after 1 warm-up, torch {torch.__version__} CPU, {ncpu} threads'}


class Watchdog:
    """Deadline per phase for the N > 1 flow: a collective that never completes (a rank that died, a communicator that cannot
    form) would otherwise end as a run killed at its time limit with nothing printed.  A daemon thread checks the deadline of
    the current phase; when it passes, it says which phase is stuck and ends the process with a non-zero code (os._exit: the
    main thread is inside a blocking call and cannot be unwound; never a re-exec)."""

    def __init__(self, rank):
        import threading
        self.rank, self.name, self.deadline = rank, 'start', None
        self._lock = threading.Lock()
        threading.Thread(target=self._run, daemon=True).start()

    def phase(self, name, seconds):
        with self._lock:
            self.name, self.deadline = name, (time.monotonic() + seconds if seconds else None)

    def _run(self):
        while True:
            time.sleep(1.0)
            with self._lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                name = self.name
            if late:
                print(f'[bench] WATCHDOG: rank {self.rank} stuck in phase "{name}" past its deadline -- giving up', file=sys.stderr, flush=True)
                os._exit(3)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run as a CHILD process,
    before this process has touched the GPU) and relay rank 0's JSON line.  The reference launches its 8 workers the same way
    (paper_scripts/*/pretrain/*.sh:8-19 -> torch.distributed.launch, pretrain.py:205-220)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] --gpus %d without WORLD_SIZE: launching %s' % (args.gpus, ' '.join(cmd[1:8])), file=sys.stderr, flush=True)
    limit = float(os.environ.get('DUALVAR_BENCH_TIMEOUT', 1500))
    # a session (= process group) of its own: on a timeout the launcher AND its N ranks are signalled -- killing only the
    # torch.distributed.run parent would orphan the ranks with their GPUs and a possibly hung collective
    import signal
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 5)):
            try:
                os.killpg(proc.pid, sig)             # (the session leader's pid is the group id)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        try:
            out = proc.stdout.read() or b''
        except Exception:
            out = b''
        sys.stdout.write(out.decode(errors='replace'))
        print(f'[bench] the {args.gpus}-rank job did not finish within {limit:.0f} s: its process group was terminated',
              file=sys.stderr, flush=True)
        sys.exit(4)
    for line in out.decode(errors='replace').splitlines():      # the JSON line to stdout, library chatter to stderr
        print(line, file=sys.stdout if line.startswith('{') else sys.stderr, flush=True)
    sys.exit(proc.returncode)


WD = None


def setup_dist(args):
    global WD
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # DUALVAR_FORCE_EXCHANGE rehearses the N > 1 code path with ONE rank: there --gpus stays 1
    if world != args.gpus:
        raise SystemExit(f'[bench] --gpus {args.gpus} but WORLD_SIZE is {world}: launch with '
                         f'`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` '
                         f'(or plain `python bench.py --gpus {args.gpus}`, which starts the ranks itself)')
    WD = Watchdog(rank)
    if args.launch_check:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        WD.phase('init_process_group(gloo)', 120)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        WD.phase('launch-check all-reduce', args.hang_check or 120)
        t = torch.ones(1)
        if args.hang_check and rank == 1:
            time.sleep(args.hang_check + 30)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({'launch_check': True, 'n_gpus': world, 'sum': float(t)}), flush=True)
        WD.phase('teardown', 60)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0)
    # DUALVAR_FORCE_EXCHANGE=1: rehearse the multi-GPU step (every RCCL collective issued) with one rank on a 1-GPU box
    distributed = world > 1 or os.environ.get('DUALVAR_FORCE_EXCHANGE') == '1'
    if os.environ.get('DUALVAR_BENCH_BACKEND'):      # rehearsal of the N > 1 flow on a 1-GPU box: ranks share the device (gloo)
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        backend = os.environ.get('DUALVAR_BENCH_BACKEND', 'nccl')
        WD.phase('init_process_group(%s)' % backend, 300)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, distributed, dev


def run_leg(args, dtype, rank, world, distributed, dev):
    """One timed leg: build the model in `dtype`, warm up, calibrate, time EXACTLY args.steps steps between barriers.
    Returns (on every rank) the result fields of that leg; rank 0's are the ones printed."""
    from dualvar_amd.optim import SGD
    from dualvar_amd.parallel import GradSync
    import dualvar_amd.engine as _eng
    wd_on = world > 1                 # deadlines only where a peer can keep us waiting
    WD.phase(f'{dtype}: build model + first steps (RCCL communicators form here)', 600 if wd_on else 0)
    torch.manual_seed(0)
    np.random.seed(1234 + rank)
    model = build_model(args, distributed)
    model.set_compute_dtype(dtype).train().to(dev)
    V = 2 if args.model.endswith('naked') else 3
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    block = torch.randn(B, V, 3, args.frames, args.size, args.size, generator=g).to(dev)
    gsync = GradSync() if distributed else None
    if gsync is not None:
        gsync.attach(model)           # bucket-wise all-reduce from inside the backward pass
    opt = SGD([p for p in model.parameters() if p.requires_grad], lr=0.003, momentum=0.9, weight_decay=1e-4,
              stores=model.stores(), grad_sync=gsync)

    def step():
        ret = model(block)
        loss = total_loss(ret)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(max(args.warmup - 1, 1)):
        step()
    # calibration step: time every launch once to find the kernel with the largest share
    # (single stream for this one step, so that every kernel's time is its own: in the timed region the weight
    # gradients run on a side stream, concurrently with the main chain -- engine.Plan.run_backward)
    WD.phase(f'{dtype}: calibration step', 300 if wd_on else 0)
    side_default = _eng.WGRAD_SIDE_STREAM
    _eng.WGRAD_SIDE_STREAM = False
    cal = KernelTimer()
    for p in all_plans(model):
        p.timer = cal
    step()
    torch.cuda.synchronize()
    _eng.WGRAD_SIDE_STREAM = side_default
    csum = cal.summary()
    if args.dump_launches and rank == 0:
        with open(args.dump_launches if dtype == args.dtype else args.dump_launches + '.' + dtype, 'w') as fh:
            fh.write('name\tkernel\tus\tGB/s\tTFLOP/s\tbytes\tflops\tshape\n')
            for l, a, b in cal.rec:
                us = a.elapsed_time(b) * 1e3
                fh.write('%s\t%s\t%.1f\t%.0f\t%.1f\t%d\t%d\t%s\n' % (l.name, l.kname, us, l.bytes / us / 1e3, l.flops / us / 1e6,
                                                                     l.bytes, l.flops, getattr(l, 'shape', '')))
    ksum = cal.summary(kernel_of)       # per GPU function: the dominant KERNEL is chosen (and timed live) at this granularity
    dominant = max((k for k in ksum if not k.startswith('host:')), key=lambda k: ksum[k][1])
    probe = KernelTimer(only=dominant)
    for p in all_plans(model):
        p.timer = probe

    WD.phase(f'{dtype}: timed region ({args.steps} steps)', (300 + 5 * args.steps) if wd_on else 0)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    from dualvar_amd import rccl as _rccl
    _rccl.STATS.reset()               # collectives issued inside the timed region (direct RCCL path; c10d calls are not counted)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    coll = _rccl.STATS.summary(args.steps) if distributed else None
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    for p in all_plans(model):
        p.timer = None
    WD.phase(f'{dtype}: report', 120 if wd_on else 0)

    print(f'[bench] {dtype}: timed region: {args.steps} steps in {dt:.3f} s', file=sys.stderr, flush=True)
    clips = world * B * V * args.steps
    n, ms, nbytes, flops = probe.summary(kernel_of)[dominant]
    avg_ms = ms / n
    bw = nbytes / n / (avg_ms * 1e-3) / 1e9              # GB/s algorithmic
    tf = flops / n / (avg_ms * 1e-3) / 1e12
    f_hbm, f_mfma = bw / HBM_PEAK_GBS, tf / MFMA_PEAK_TF[dtype]
    if f_mfma > f_hbm:
        roof = {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': MFMA_PEAK_TF[dtype], 'unit': 'TFLOP/s', 'frac': round(f_mfma, 4)}
    else:
        roof = {'bound': 'hbm', 'achieved': round(bw, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(f_hbm, 4)}
    tot_ms = sum(v[1] for k, v in csum.items() if not k.startswith('host:'))
    traffic, roof_src = None, None
    try:        # HBM bytes per launch of this kernel family from the committed PMC passes (tools/pmc_traffic.py)
        for pj in sorted((f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_pmc_traffic.json')), reverse=True):
            fam = json.load(open(os.path.join(ROOT, 'profiles', pj)))['families']
            # (conv_tap's forward and data gradient are one kernel: the PMC tables have one family for both)
            key = dominant.replace('+bn_in', '')         # (the BatchNorm-on-load instantiations are counted with their family)
            key = key if key in fam else key.replace('FWD,', '').replace('DGRAD,', '').replace('+bn_reduce', '')
            if key in fam:
                traffic, roof_src = round(fam[key]['hbm_bytes_per_launch']), pj
                break
    except Exception:
        traffic = None
    # the same kernel alone on the GPU (calibration step, one stream): what the kernel itself achieves
    cn, cms, cbytes, cflops = ksum[dominant]
    iso_bw, iso_tf = cbytes / (cms * 1e-3) / 1e9, cflops / (cms * 1e-3) / 1e12
    roof['isolated'] = {'avg_launch_us': round(cms / cn * 1e3, 2), 'GB/s': round(iso_bw, 1), 'TFLOP/s': round(iso_tf, 2),
                        'frac_hbm': round(iso_bw / HBM_PEAK_GBS, 4), 'frac_mfma': round(iso_tf / MFMA_PEAK_TF[dtype], 4)}
    roof['concurrent'] = bool(side_default and dominant.startswith('conv_wgrad'))
    if dtype == 'fp32' and not _dv_lib.f32_exact():
        roof['split_peak'] = round(F32_SPLIT_PEAK_TF, 1)
        roof['frac_of_split_peak'] = round(tf / F32_SPLIT_PEAK_TF, 4)
        roof['note'] = ('fp32 products = 6 bf16 MFMAs per 32x32x16 block (exact 3-way bf16 split of both operands): `peak` is the '
                        'fp32 matrix peak of the guide, `split_peak` the bf16 dense peak / 6')
    roof.update({'traffic': traffic, 'kernel': dominant, 'launches_per_step': n // args.steps,
                 'avg_launch_us': round(avg_ms * 1e3, 2), 'share_of_kernel_time': round(ksum[dominant][1] / tot_ms, 3),
                 'other_bound_frac': round(min(f_hbm, f_mfma), 4),
                 'algorithmic_bytes_per_launch': round(nbytes / n),
                 'algorithmic_flops_per_launch': round(flops / n),
                 'traffic_source': (roof_src + ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)') if traffic else None})
    # whole step against both roofs (algorithmic bytes / flops of every plan launch)
    pb = sum(l.bytes for p in all_plans(model) for l in p.f_list + p.b_list)
    pf = sum(l.flops for p in all_plans(model) for l in p.f_list + p.b_list)
    step_s = dt / args.steps
    res = {'dtype': dtype, 'value': round(clips / dt, 2), 'ms_per_step': round(step_s * 1e3, 3), 'loss': round(float(loss.detach()), 4),
           'roofline': roof,
           'whole_step': {'algorithmic_GB': round(pb / 1e9, 3), 'algorithmic_TFLOP': round(pf / 1e12, 4),
                          'frac_hbm': round(pb / step_s / 1e9 / HBM_PEAK_GBS, 4),
                          'frac_mfma': round(pf / step_s / 1e12 / MFMA_PEAK_TF[dtype], 4)},
           'kernel_time_ms_per_step': {k: round(v[1], 3) for k, v in sorted(csum.items(), key=lambda kv: -kv[1][1])[:12]}}
    if coll is not None:
        # per step and rank: how many collectives the step enqueues in-stream, their payload and the HOST time of the enqueue calls
        # (the one-rank rehearsal under DUALVAR_FORCE_EXCHANGE=1 gives the code-path cost; a real N-GPU line adds wire latency)
        res['collectives'] = {k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in coll.items()}
    del model, opt, block
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)             # never returns; nothing in this process has touched the GPU yet
    rank, world, distributed, dev = setup_dist(args)
    V = 2 if args.model.endswith('naked') else 3
    B = args.batch
    first = run_leg(args, args.dtype, rank, world, distributed, dev)
    sec = args.secondary
    if sec == 'auto':
        sec = 'bf16' if args.dtype == 'fp32' else 'none'
    second = run_leg(args, sec, rank, world, distributed, dev) if sec not in ('none', args.dtype) else None

    if rank == 0:
        out = {
            'metric': ('clips/sec (whole node), S3D-G 8x112^2 SimCLR pretrain step'
                       if (args.net, args.model, args.frames, args.size) == ('s3dg', 'simclr_naked', 8, 112) else
                       f'clips/sec (whole node), {args.net} {args.frames}x{args.size}^2 {args.model} pretrain step'),
            'value': first['value'],
            'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': first['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': first['dtype'], 'data': 'synthetic',
            'config': {'workload': f'{args.net} {args.model} pretrain step (fwd+loss+bwd+SGD), {args.frames}x{args.size}x{args.size} '
                                   f'RGB clips, {B} samples x {V} views per GPU, random-init weights',
                       'global_batch': world * B, 'clips_per_step': world * B * V, 'parallelism': f'dp{world}',
                       'arithmetic': ('fp32 storage, fp32 accumulate; products on the bf16 matrix cores through an exact 3-way bf16 split of '
                                      'each fp32 operand (6 partial products, error at fp32 rounding level; DUALVAR_F32_EXACT=1: exact-f32 '
                                      'MFMA).  The reference\'s arithmetic: parity with the reference within 1e-3 on the loss is asserted '
                                      'in this mode (tests/test_models_gpu.py)')
                       if first['dtype'] == 'fp32' else
                       ('bf16 storage, fp32 accumulate / statistics' if first['dtype'] == 'bf16' else
                        'bf16 storage; bottleneck 1x1x1 convs: e4m3 x e4m3 (forward) / e5m2 x e4m3 (data gradient) on the fp8 MFMA, fp32 '
                        'accumulate, per-tensor scales')},
            'loss': first['loss'],
            'roofline': first['roofline'],
            'whole_step': first['whole_step'],
            'kernel_time_ms_per_step': first['kernel_time_ms_per_step'],
        }
        if 'collectives' in first:
            out['extra'] = {'collectives_per_step_and_rank': first['collectives'],
                            'note': 'in-stream RCCL calls inside the timed region (rank 0): count, payload bytes and HOST time of the '
                                    'enqueue calls per step; with one rank under DUALVAR_FORCE_EXCHANGE=1 this is the code-path cost only'}
        if second is not None:
            out['secondary'] = dict(second, note=(
                'same step with bf16 STORAGE (fp32 accumulate and statistics): throughput mode.  Its outputs are bounded against '
                'the reference by what bf16 rounding of every activation does to the reference itself '
                '(tests/test_models_gpu.py::test_backbone_features[bf16]), not by the 1e-3 north-star tolerance -- '
                'hence not the headline value.') if second['dtype'] == 'bf16' else 'same step in fp32')
        if not args.no_cpu_baseline and world == 1:          # the host-core baseline is reported at N=1 only
            out['cpu_baseline'] = cpu_baseline(args, V)
        print(json.dumps(out))
    if distributed:
        WD.phase('teardown', 120 if world > 1 else 0)
        dist.barrier()
        torch.cuda.synchronize()
        from dualvar_amd import rccl
        rccl.destroy_all()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
