import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train(); g.precision = "f32"
inp = synthetic.make_inputs(h, 32, 256, seed=4, device=dev)
for rep in range(3):
    el, ms = bench.run_steps(g, inp, 60, 8)
    print(f'wall {el / 60 * 1e3:.3f} ms/step; events: min {min(ms):.3f} median {sorted(ms)[30]:.3f}')
per = {}
with torch.no_grad():
    for it in range(5):
        g._profile = []
        g(*inp); torch.cuda.synchronize()
        if it >= 2:
            for tag, e0, e1 in g._profile: per.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3)
print({k[-12:]: round(sum(v) / len(v)) for k, v in per.items()})
