"""Time of conv_pre's own weight fold (split_rowscale + split_pack of one 512 x 768 x 7 layer: on the forward's critical path) and of the batch of
the other Conv1d layers.  tools/exp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import hipops
dev = torch.device('cuda:0')
def plan(layers):
    subs = []
    for co, ci, k in layers:
        v = torch.randn(co, ci, k, device=dev) * 0.02
        g = torch.rand(co, 1, 1, device=dev) + 0.5
        frag = torch.empty(hipops.split_halves(k, ci, co) + 1024, device=dev, dtype=torch.float16)
        subs.append((v, g, frag, torch.empty(4, device=dev)))
    return hipops.SplitPlan(subs, dev, bf16=True), subs
for name, layers in (('conv_pre', [(512, 768, 7)]), ('the 30 residual convs', [(c, c, k) for c in (256, 128, 64) for k in (3, 7, 11) for _ in range(2)] )):
    p, subs = plan(layers)
    for _ in range(5):
        p.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            p.run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    chk = float(subs[0][2].view(torch.bfloat16)[:4096].float().abs().sum())
    print(f'{name}: {best * 1e3:.1f} us per fold (two launches back to back)   checksum {chk:.6f}', flush=True)
