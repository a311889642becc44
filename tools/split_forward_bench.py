#!/usr/bin/env python3
"""cfg2 forward: exact-fp32 path vs precision='f16x3' (time per forward and output difference)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavthruvec_pytorch_amd import Generator, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
h = synthetic.make_hparams(num_wv_feat=768)
inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
ys = {}
for prec in ('f32', 'f16x3', 'bf16'):
    g = Generator(h); g.load_state_dict(synthetic.make_state_dict(h, seed=0)); g = g.to(dev).train()
    g.precision = prec
    with torch.no_grad():
        for _ in range(3): y = g(*inp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = g(*inp)
        e1.record(); torch.cuda.synchronize()
    ys[prec] = y.clone()
    ms = e0.elapsed_time(e1) / 10
    print(f'{prec:6s}: {ms:7.3f} ms/forward  {B * T * 320 / ms / 1e3:8.1f} M samples/s')
print('max|y_f16x3 - y_f32| = %.3g   max|y_bf16 - y_f32| = %.3g   |y|max = %.3g' % (
    (ys['f32'] - ys['f16x3']).abs().max().item(), (ys['f32'] - ys['bf16']).abs().max().item(), ys['f32'].abs().max().item()))
