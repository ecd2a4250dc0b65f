"""TEST-INFRASTRUCTURE probe (CPU, uses the oracle): how well conditioned is a parity case?

For S3D-G SimCLR_Naked at several batch sizes / initialisations it reports, for the oracle itself,
  * fp32 vs fp64 (loss, logits, pooled)            -- the floor any fp32 implementation sees
  * emulated bf16 storage vs fp32 (loss, logits)   -- what bf16 storage rounding alone does
so that a fixture can be chosen on which a bf16 implementation CAN meet the north-star 1e-3 loss bound.

    python tools/conditioning_probe.py [B ...]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import procedural as P, torch_ref as O   # noqa: E402


def bf16_hooks(model):
    return [mod.register_forward_hook(lambda _m, _i, out: out.to(torch.bfloat16).float()) for mod in model.modules()
            if isinstance(mod, (torch.nn.Conv3d, torch.nn.BatchNorm3d, torch.nn.MaxPool3d))]


def run(model, block):
    ret = model(block)
    return float(ret['clip_contrast_loss']), ret['clip_logits'].detach().double().numpy()


def main():
    torch.set_num_threads(8)
    net = os.environ.get('NET', 's3dg')
    T = int(os.environ.get('FRAMES', 8))
    H = int(os.environ.get('SIZE', 112))
    gain = float(os.environ.get('GAIN', 1.0))
    inp = os.environ.get('INPUT', 'procedural')
    for B in [int(a) for a in sys.argv[1:]] or [4, 8, 16]:
        if inp == 'procedural':
            block = P.procedural_clips(B, 2, T, H, H)
        else:
            block = torch.randn(B, 2, 3, T, H, H, generator=torch.Generator().manual_seed(1234))
        t0 = time.time()
        torch.manual_seed(0)
        m = O.SimCLR_Naked(net, 128, 0.07, False)
        P.procedural_init(m, gain=gain)
        m.train()
        with torch.no_grad():
            l32, g32 = run(m, block)
            hooks = bf16_hooks(m)
            lb, gb = run(m, block.to(torch.bfloat16).float())
            for h in hooks:
                h.remove()
            m64 = O.SimCLR_Naked(net, 128, 0.07, False)
            P.procedural_init(m64, gain=gain)
            m64.train().double()
            l64, g64 = run(m64, block.double())
        print(f'{net} B={B} ({2 * B} clips {T}x{H}^2, input {inp}, gain {gain}): loss {l64:.6f}  '
              f'fp32-fp64: dloss {abs(l32 - l64):.2e} dlogits {np.abs(g32 - g64).max():.2e}   '
              f'bf16emu-fp32: dloss {abs(lb - l32):.2e} dlogits {np.abs(gb - g32).max():.2e}  '
              f'[{time.time() - t0:.0f} s]', flush=True)


if __name__ == '__main__':
    main()
