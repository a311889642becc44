"""cond_stream on / off over the forward's size (bf16 train-mode forward, in-process A/B): where does the second side stream pay?  tools/exp."""
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
for B, T in ((8, 256), (16, 256), (32, 256), (48, 256), (64, 256), (32, 512), (96, 256), (64, 512)):
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    res = {}
    for rep in range(2):
        for cs in (False, True):
            g.cond_stream = cs
            with torch.no_grad():
                for _ in range(4):
                    g(*inp)
                best = 1e9
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(); e0.record()
                    for _ in range(40):
                        g(*inp)
                    e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 40)
            res[cs] = min(res.get(cs, 1e9), best)
    print(f'B={B:3d} T={T:4d} B*T={B * T:6d}: off {res[False] * 1e3:8.1f} us   on {res[True] * 1e3:8.1f} us   ({(res[True] - res[False]) * 1e3:+.1f})', flush=True)
