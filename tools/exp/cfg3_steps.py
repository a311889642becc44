import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train(); g.precision = 'bf16'
inp = synthetic.make_inputs(h, 64, 512, seed=4, device=dev)
for rep in range(3):
    el, ms = bench.run_steps(g, inp, 40, 3)
    print(f'wall {el / 40 * 1e3:.3f} ms/step; events: min {min(ms):.3f} median {sorted(ms)[20]:.3f} mean {sum(ms) / 40:.3f} max {max(ms):.3f}')
    print(' '.join(f'{m:.2f}' for m in ms))
# host time of one forward (no sync)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(10):
        g(*inp)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f'host issue time {(t1 - t0) / 10 * 1e3:.3f} ms/forward')
