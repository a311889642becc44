"""Eval-mode forward time, f32 path against bf16 path, over small B x T (where does the bf16 path start to pay?).  tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h)
g.load_state_dict(synthetic.make_state_dict(h, seed=0))
g = g.to(dev).eval()
for B, T in ((1, 50), (1, 100), (1, 200), (1, 400), (2, 400), (4, 256), (4, 400), (8, 256), (16, 256), (32, 256)):
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    res = {}
    for prec in ('f32', 'bf16'):
        g.precision = prec
        with torch.no_grad():
            for _ in range(4):
                g(*inp)
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(20):
                    g(*inp)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
        res[prec] = best
    print(f'B={B:3d} T={T:4d} B*T={B * T:6d}: f32 {res["f32"] * 1e3:8.1f} us   bf16 {res["bf16"] * 1e3:8.1f} us', flush=True)
