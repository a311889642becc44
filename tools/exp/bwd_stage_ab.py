"""A/B on one box: the narrow stages' input gradients as one kernel per stage (Generator.fuse_stage_backward) or as three merged launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
B, T = 32, 256
inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
dy = torch.randn(B, 1, T * 320, device=dev)
gs = {}
for on in (True, False):
    g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train(); g.precision = os.environ.get('V2W_AB_PREC', 'f32')
    setattr(g, os.environ.get("V2W_AB_ATTR", "fuse_stage_backward"), on)
    gs[on] = (g, torch.optim.AdamW(g.parameters(), 2e-4, betas=(0.8, 0.99)))
for rep in range(3):
    for on in (True, False):
        g, opt = gs[on]
        for it in range(12):
            if it == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            (g(*inp) * dy).sum().backward()
            opt.step()
        torch.cuda.synchronize()
        print(f'{os.environ.get("V2W_AB_ATTR", "fuse_stage_backward")}={on}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step')
