#!/usr/bin/env python3
"""Time v2w_conv_post_tanh_bf16in at the BASELINE configs[2] shape for variant libraries (tools/res_timeline.py build VARIANT DEFS...)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if sys.argv[1] == 'child':
    v = sys.argv[2]
    if v != 'main':
        os.environ['V2W_LIB'] = os.path.join(ROOT, 'tools', 'exp', f'libv2w_res{v}.so')
    import torch
    from wavthruvec_pytorch_amd import hipops
    dev = torch.device('cuda:0')
    B, C, L = 64, 16, 163840
    x = torch.randn(B, C, L, device=dev).bfloat16()
    w = torch.randn(7, C, 1, device=dev) / 10
    b = torch.zeros(1, device=dev)
    out = torch.empty(B, 1, L, device=dev)
    run = lambda: hipops.conv_post_tanh(x, w, b, out, k=7, slope=0.01)
    for _ in range(5): run()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f'[{v}] conv_post {ts[len(ts)//2]:.1f} us  ({(B*C*L*2 + B*L*4) / ts[len(ts)//2] / 1e6:.2f} TB/s)', flush=True)
else:
    for rnd in range(2):
        for v in sys.argv[1:]:
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child', v])
