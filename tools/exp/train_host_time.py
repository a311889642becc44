#!/usr/bin/env python3
"""Host-side issue time of one training step (forward / backward / AdamW) against its GPU time: is the step launch-bound anywhere?
argv: B T [profile]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from wavthruvec_pytorch_amd import Generator, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h)
g.load_state_dict(synthetic.make_state_dict(h, seed=0))
g = g.to(dev).train()
opt = torch.optim.AdamW(g.parameters(), 2e-4, betas=(0.8, 0.99))
inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
dy = torch.randn(B, 1, T * 320, device=dev)


def step(times=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    y = g(*inp)
    t1 = time.perf_counter()
    (y * dy).sum().backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if times is not None:
        times.append((t1 - t0, t2 - t1, t3 - t2))


for _ in range(4):
    step()
torch.cuda.synchronize()
for rep in range(3):
    ts = []
    t0 = time.perf_counter()
    for _ in range(5):
        step(ts)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tg = time.perf_counter() - t0
    f = sum(t[0] for t in ts) / 5e-3; b = sum(t[1] for t in ts) / 5e-3; o = sum(t[2] for t in ts) / 5e-3
    print(f'5 steps: host issue {th / 5e-3:.2f} ms/step (forward {f:.2f}, backward {b:.2f}, AdamW {o:.2f}); with the GPU drained {tg / 5e-3:.2f} ms/step')
    print('   first step of the burst (queue empty): forward %.2f backward %.2f AdamW %.2f ms' % tuple(1e3 * v for v in ts[0]))
if len(sys.argv) > 3:
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(35)
