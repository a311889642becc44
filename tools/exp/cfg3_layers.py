#!/usr/bin/env python3
"""Per-launch event times of the configs[2] bf16 forward for variant libraries: python tools/exp/cfg3_layers.py main _variant ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if sys.argv[1] == 'child':
    v = sys.argv[2]
    flags = v.split(':')[1:]          # 'main:nopost', '_variant:noup': Generator switches
    v = v.split(':')[0]
    if v != 'main':
        os.environ['V2W_LIB'] = os.path.join(ROOT, 'tools', 'exp', f'libv2w_res{v}.so')
    import torch
    from wavthruvec_pytorch_amd import Generator, synthetic
    dev = torch.device('cuda:0')
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train(); g.precision = 'bf16'
    if 'nopost' in flags: g.fuse_post = False
    if 'noup' in flags: g.fuse_up = False
    v = ':'.join([v] + flags)
    inp = tuple(t.to(dev) for t in synthetic.make_inputs(h, 64, 512, seed=1))
    per = {}
    with torch.no_grad():
        for it in range(6):
            g._profile = []
            g(*inp); torch.cuda.synchronize()
            if it >= 2:
                for tag, e0, e1 in g._profile: per.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3)
    g._profile = None
    tot = 0
    out = []
    for tag, ts in per.items():
        ts.sort(); m = ts[len(ts) // 2]; tot += m
        if m > 20: out.append(f'{tag.split(":")[-1][:28]}={m:.0f}')
    print(f'[{v}] sum {tot:.0f} us: ' + ' '.join(out), flush=True)
else:
    for rnd in range(2):
        for v in sys.argv[1:]:
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child', v])
