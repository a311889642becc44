"""The no-grad train-mode forwards that follow training steps (tools/train_step_bench.py's second figure), one by one.  tools/exp."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
g = Generator(h)
g.load_state_dict(synthetic.make_state_dict(h, seed=0))
g = g.to(dev).train()
opt = torch.optim.AdamW(g.parameters(), 2e-4, betas=(0.8, 0.99))
inp = synthetic.make_inputs(h, 32, 256, seed=1, device=dev)
dy = torch.randn(32, 1, 256 * 320, device=dev)
for it in range(3):
    opt.zero_grad(set_to_none=True)
    (g(*inp) * dy).sum().backward()
    opt.step()
torch.cuda.synchronize()
with torch.no_grad():
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g(*inp); torch.cuda.synchronize()
        print(f'no-grad forward {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms  tapes {len(g._tapes)} refused {g._tape_refused}', flush=True)
