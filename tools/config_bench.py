#!/usr/bin/env python3
"""Forward time (event-timed median of 10 steps) of the BASELINE.json configurations that fit one GPU, in the three precision modes
(train-mode CondBN)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic

dev = torch.device('cuda:0')
CFGS = {
    'cfg1 (B=1, T=50, 768-d, x320)': (dict(num_wv_feat=768), 1, 50),
    'cfg2 (B=32, T=256, 768-d, x320)': (dict(num_wv_feat=768), 32, 256),
    'cfg3 (B=64, T=512, 768-d, x320)': (dict(num_wv_feat=768), 64, 512),
    'cfg5 (B=16, T=256, 1024-d, x640)': (dict(num_wv_feat=1024, upsample_rates=[8, 5, 4, 2, 2], upsample_kernel_sizes=[16, 11, 8, 4, 4]), 16, 256),
}
for name, (hp, B, T) in CFGS.items():
    h = synthetic.make_hparams(**hp)
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    up = synthetic.total_upsample(h)
    out, ys = [], {}
    for prec in ('f32', 'f16x3', 'bf16'):
        g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train(); g.precision = prec
        with torch.no_grad():
            for _ in range(3): y = g(*inp)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            evs[0].record()
            for i in range(10):
                y = g(*inp)
                evs[i + 1].record()
            torch.cuda.synchronize()
        ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(10))[5]       # median step (an allocator stall in one step is not the kernels' time)
        ys[prec] = y
        out.append(f'{prec} {ms:7.3f} ms ({B * T * up / ms / 1e3:6.1f} M/s)')
        del g
    print(f'{name:36s} ' + '   '.join(out) + f'   |f16x3-f32| {(ys["f16x3"] - ys["f32"]).abs().max().item():.1e}  |bf16-f32| {(ys["bf16"] - ys["f32"]).abs().max().item():.1e}')
