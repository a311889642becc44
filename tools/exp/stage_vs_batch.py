"""Per-launch times of the bf16 train-mode forward over the batch size at T=256 (does a stage kernel's time step with ceil(workgroups / slots)?).
tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h)
g.load_state_dict(synthetic.make_state_dict(h, seed=0))
g = g.to(dev).train()
g.precision = 'bf16'
for B in (24, 26, 28, 30, 32, 34, 36, 40, 48):
    inp = synthetic.make_inputs(h, B, 256, seed=1, device=dev)
    acc = {}
    with torch.no_grad():
        for _ in range(4):
            g(*inp)
        for _ in range(10):
            g._profile = []
            g(*inp)
            torch.cuda.synchronize()
            for tag, e0, e1 in g._profile:
                acc.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3)
            g._profile = None
    row = '  '.join(f'{min(v):6.1f}' for v in acc.values())
    per = '  '.join(f'{min(v) / B:6.2f}' for v in acc.values())
    print(f'B={B:3d}: us {row}   | us per item {per}', flush=True)
print('columns:', list(acc.keys()))
