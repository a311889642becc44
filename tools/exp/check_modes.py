import sys, os
sys.path.insert(0, '/root/repo')
import torch
from wavthruvec_pytorch_amd import Generator, synthetic
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
sd = synthetic.make_state_dict(h, seed=0)
# 1. graph capture in f16x3 eval mode
g = Generator(h); g.load_state_dict(sd); g = g.to(dev).eval(); g.precision = 'f16x3'
inp = synthetic.make_inputs(h, 1, 50, seed=3, device=dev)
with torch.no_grad():
    y0 = g(*inp).clone()
run = g.capture_graph(*inp)
y1 = run(*inp).clone()
print('graph f16x3 eval: max diff', (y0 - y1).abs().max().item())
# 2. cfg3 shape in bf16 mode (B=64, T=512)
g2 = Generator(h); g2.load_state_dict(sd); g2 = g2.to(dev).train(); g2.precision = 'bf16'
inp3 = synthetic.make_inputs(h, 64, 512, seed=4, device=dev)
with torch.no_grad():
    for _ in range(2): y = g2(*inp3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = g2(*inp3)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f'cfg3 (B=64, T=512, bf16 operands): {ms:.2f} ms/forward = {64 * 512 * 320 / ms / 1e3:.0f} M samples/s; finite={torch.isfinite(y).all().item()} |y|max={y.abs().max().item():.3f}')
g2.precision = 'f32'
with torch.no_grad():
    y32 = g2(*inp3)
    g2.precision = 'bf16'
    yb = g2(*inp3)
print('cfg3 max|y_bf16 - y_f32| =', (y32 - yb).abs().max().item())
print('mem GB', torch.cuda.max_memory_allocated() / 1e9)
