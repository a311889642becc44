#!/usr/bin/env python3
"""Gradient deviation of a bf16-arithmetic training step from the fp32 oracle, beside the reference's own autocast backward (the oracle's modules
under torch.autocast(bfloat16)) on the same inputs.  argv: B T"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import vec2wav_oracle as O  # noqa: E402
from wavthruvec_pytorch_amd import Generator, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768, resblock='1' if os.environ.get('RB1') else 1)
sd = synthetic.make_state_dict(h, seed=0)
inp = synthetic.make_inputs(h, B, T, seed=21)
dy = torch.from_numpy(np.random.default_rng(5).standard_normal((B, 1, T * 320)).astype(np.float32))
_, g_ref, _ = O.generator_gradients(sd, h, *inp, dy, training=True)
with torch.autocast('cpu', dtype=torch.bfloat16):
    _, g_ac, _ = O.generator_gradients(sd, h, *inp, dy, training=True)


def run(precision, **kw):
    g = Generator(h)
    g.load_state_dict(sd)
    g = g.to(dev).train()
    g.precision = precision
    for k, v in kw.items():
        setattr(g, k, v)
    y = g(*[t.to(dev) for t in inp])
    (y * dy.to(dev)).sum().backward()
    return {n: p.grad.cpu() for n, p in g.named_parameters()}


def dev_of(ga):
    out = {}
    for n, ref in g_ref.items():
        floor = 0.25 if (n.startswith('ups.') and n.endswith('.bias')) else 1e-6
        sc = max(ref.abs().max().item(), floor)
        e = (ga[n].float() - ref)
        out[n] = (e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / max(ref.pow(2).mean().sqrt().item(), floor))
    return out


d_ac = dev_of(g_ac)
runs = {'bf16': dev_of(run('bf16'))}
if len(sys.argv) > 3:
    runs['bf16+train_storage'] = dev_of(run('bf16', bf16_training=True))
for name, d in runs.items():
    worse = [(n, d[n], d_ac[n]) for n in d if d[n][0] > d_ac[n][0]]
    print(f'== {name}: {len(worse)} of {len(d)} parameters farther (max) from the fp32 oracle than the reference autocast')
    print('   worst relative-to-autocast:')
    for n, a, b in sorted(worse, key=lambda t: -t[1][0] / max(t[2][0], 1e-12))[:12]:
        print(f'     {n:42s} hip max {a[0]:.2e} rms {a[1]:.2e} | autocast max {b[0]:.2e} rms {b[1]:.2e}')
    rat = sorted(d[n][0] / max(d_ac[n][0], 1e-9) for n in d)
    print('   ratio hip/autocast (max dev) quantiles: min %.2f 25%% %.2f 50%% %.2f 75%% %.2f 90%% %.2f max %.2f' % (rat[0], rat[len(rat)//4], rat[len(rat)//2], rat[3*len(rat)//4], rat[9*len(rat)//10], rat[-1]))
    print('   overall: hip max-of-max %.2e mean-of-max %.2e | autocast max-of-max %.2e mean-of-max %.2e' % (
        max(v[0] for v in d.values()), np.mean([v[0] for v in d.values()]), max(v[0] for v in d_ac.values()), np.mean([v[0] for v in d_ac.values()])))
    print('   overall rms: hip max %.2e mean %.2e | autocast max %.2e mean %.2e' % (
        max(v[1] for v in d.values()), np.mean([v[1] for v in d.values()]), max(v[1] for v in d_ac.values()), np.mean([v[1] for v in d_ac.values()])))
