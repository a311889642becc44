#!/usr/bin/env python3
"""Time the fused narrow-stage kernels at cfg2 shapes: fp32 (resblock2_stage) vs split f16x3 / bf16 (resblock2_stage_split)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavthruvec_pytorch_amd import hipops

dev = torch.device('cuda:0')
for C, L in ((32, 40960), (16, 81920)):
    B = 32
    x = torch.randn(B, C, L, device=dev); a = torch.ones(B, C, device=dev); s = torch.zeros(B, C, device=dev)
    out = torch.empty_like(x)
    br32, brs, brb = [], [], []
    ks = (3, 7, 11)
    tot = sum(2 * hipops.split_units_halves(k, C, C) for k in ks) + 1024
    ar_s = torch.zeros((tot,), device=dev, dtype=torch.float16); ar_b = torch.zeros((tot,), device=dev, dtype=torch.float16)
    off = 0
    for k in ks:
        ws = [torch.randn(k, C, C, device=dev) / (C * k) ** 0.5 for _ in range(2)]
        bs = [torch.zeros(C, device=dev) for _ in range(2)]
        n = hipops.split_units_halves(k, C, C)
        ps = [hipops.pack_split(ws[i], out=ar_s[off + i * n: off + (i + 1) * n]) for i in range(2)]
        pb = [hipops.pack_split(ws[i], out=ar_b[off + i * n: off + (i + 1) * n], bf16=True) for i in range(2)]
        off += 2 * n
        br32.append(dict(wp1=hipops.pack_mfma(ws[0]), b1=bs[0], wp2=hipops.pack_mfma(ws[1]), b2=bs[1], k=k, dil1=1, dil2=3))
        brs.append(dict(wps1=ps[0], b1=bs[0], wps2=ps[1], b2=bs[1], k=k, dil1=1, dil2=3))
        brb.append(dict(wps1=pb[0], b1=bs[0], wps2=pb[1], b2=bs[1], k=k, dil1=1, dil2=3))
    runs = {'f32': lambda: hipops.resblock2_stage(x, (a, s), br32, out, slope=0.1, out_div=3.0),
            'f16x3': lambda: hipops.resblock2_stage_split(x, (a, s), brs, out, slope=0.1, out_div=3.0),
            'bf16': lambda: hipops.resblock2_stage_split(x, (a, s), brb, out, slope=0.1, out_div=3.0, bf16=True)}
    res = []
    for name, fn in runs.items():
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(f'{name} {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us')
    print(f'C={C} L={L}: ' + '   '.join(res))
