#!/usr/bin/env python3
"""Forward time of the six-stage x640 generator (8-channel last stage), per launch group: python tools/exp/six_stage_time.py [B T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 256)
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768, upsample_rates=[5, 4, 4, 2, 2, 2], upsample_kernel_sizes=[11, 8, 8, 4, 4, 4])
for prec in ('f32', 'bf16'):
    g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train(); g.precision = prec
    inp = tuple(t.to(dev) for t in synthetic.make_inputs(h, B, T, seed=1))
    per = {}
    with torch.no_grad():
        for it in range(5):
            g._profile = []
            g(*inp); torch.cuda.synchronize()
            if it >= 2:
                for tag, e0, e1 in g._profile: per.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3)
    g._profile = None
    tot = 0; out = []
    for tag, ts in per.items():
        ts.sort(); m = ts[len(ts) // 2]; tot += m
        out.append(f'{tag.split(":")[-1][:22]}={m:.0f}')
    print(f'[{prec}] B={B} T={T} sum {tot:.0f} us: ' + ' '.join(out), flush=True)
