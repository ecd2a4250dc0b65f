"""GPU: the drop-in entry script end to end (reference pretrain.py:90-164 CLI, :394-466 loop) -- a few iterations of each
data path: float clips (`--dataset synthetic`) and decoded uint8 frames augmented inside the ingest kernel
(`--dataset synthetic-frames`)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('extra', [['--dataset', 'synthetic', '--model', 'simclr_timeseriesv4', '--num_seq', '3'],
                                   ['--dataset', 'synthetic-frames', '--model', 'simclr_naked', '--num_seq', '2', '--rand_flip',
                                    '--aug_temp_consist'],
                                   ['--dataset', 'synthetic', '--model', 'moco_naked', '--num_seq', '2', '--moco-k', '1024',
                                    '--dtype', 'fp32'],
                                   # uint8 frames through the dual-head objectives (segment shuffle on a FrameBatch, and
                                   # MoCo's [aug ; aug] pass = FrameBatch.cat)
                                   ['--dataset', 'synthetic-frames', '--model', 'simclr_timeseriesv4', '--num_seq', '3'],
                                   ['--dataset', 'synthetic-frames', '--model', 'moco_timeseriesv4', '--num_seq', '3',
                                    '--moco-k', '1024']])
def test_pretrain_script_runs(gpu, extra, tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, 'pretrain.py'), '--net', 'r3d', '--batch_size', '8', '--seq_len', '8', '--img_dim', '64',
           '--steps', '4', '--epochs', '1', '--epoch_size', '64', '--print_freq', '1', '-j', '0', '--prefix', 't'] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    m = re.search(r'clips/s \(whole job\):([0-9.]+)', out)
    assert m and float(m.group(1)) > 0, out[-2000:]
    losses = [float(v) for v in re.findall(r'VLoss ([0-9.]+)', out)]
    assert losses and all(0.0 < v < 50.0 for v in losses), losses


def test_pretrain_input_pipeline_with_workers_and_prefetch(gpu, tmp_path):
    """The steady-state input path (reference pretrain.py:394-401,550-564: DataLoader workers -> pinned memory -> .cuda()):
    uint8 frames and the augmentation rows drawn in two DataLoader workers, copied to the GPU one step ahead on a side stream,
    meters read one step late; the epoch reports a steady-state clips/s after the warm-up steps."""
    cmd = [sys.executable, os.path.join(ROOT, 'pretrain.py'), '--net', 'r3d', '--batch_size', '8', '--seq_len', '8', '--img_dim', '64',
           '--steps', '10', '--warm_steps', '3', '--epochs', '1', '--epoch_size', '160', '--print_freq', '2', '-j', '2', '--prefix', 't',
           '--dataset', 'synthetic-frames', '--model', 'simclr_naked', '--num_seq', '2', '--rand_flip']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    m = re.search(r'steady-state clips/s \(whole job, steps 2\.\.10\):([0-9.]+)', out)
    assert m and float(m.group(1)) > 0, out[-2000:]
    losses = [float(v) for v in re.findall(r'VLoss ([0-9.]+)', out)]
    assert losses and all(0.0 < v < 50.0 for v in losses), losses
