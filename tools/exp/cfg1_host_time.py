"""Host issue time of one eval-mode B = 1 x T = 50 forward (a replayed launch plan) against its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).eval()
inp = synthetic.make_inputs(h, 1, 50, seed=1, device=dev)
with torch.no_grad():
    for _ in range(20):
        g(*inp)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            g(*inp)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        print(f'200 forwards: host issue {th / 200 * 1e6:.1f} us per forward, with the GPU drained {tg / 200 * 1e6:.1f} us per forward')
