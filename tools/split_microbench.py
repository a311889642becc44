#!/usr/bin/env python3
"""Time V2W_ALGO_SPLIT against V2W_ALGO_MFMA on the generator's wide layers (cfg2 shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavthruvec_pytorch_amd import hipops

dev = torch.device('cuda:0')
CASES = [(32, 768, 512, 256, 7, 1), (32, 256, 256, 1280, 11, 1), (32, 256, 256, 1280, 3, 3), (32, 128, 128, 5120, 7, 1),
         (32, 128, 128, 5120, 11, 3), (32, 64, 64, 20480, 7, 1), (32, 64, 64, 20480, 3, 1),
         (32, 256, 256, 1024, 11, 1), (16, 256, 256, 1024, 11, 1), (8, 256, 256, 1024, 11, 1)]   # 512 / 256 / 128 workgroups of 128 x 128
if len(sys.argv) > 1:
    CASES = [CASES[int(a)] for a in sys.argv[1:]]
for B, ci, co, L, k, d in CASES:
    x = torch.randn(B, ci, L, device=dev)
    wf = torch.randn(k, ci, co, device=dev) / (ci * k) ** 0.5
    out = torch.empty(B, co, L, device=dev)
    BF = os.environ.get('V2W_MB_BF16') == '1'
    SPL = hipops.ALGO_BF16 if BF else hipops.ALGO_SPLIT
    wp, wps = hipops.pack_mfma(wf), hipops.pack_split(wf, bf16=BF)
    res = x if ci == co else None
    def run(algo):
        kw = dict(k=k, dil=d, slope=0.1, res=res, algo=algo)
        if algo == SPL: kw['wps'] = wps
        else: kw['wp'] = wp
        hipops.conv1d(x, None, None, out, **kw)
    ts = {}
    for name, algo in (('f32', hipops.ALGO_MFMA), ('split', SPL)):
        for _ in range(3): run(algo)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(algo)
        e1.record(); torch.cuda.synchronize()
        ts[name] = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * L * ci * co * k
    byt = 4.0 * B * L * (ci + co * (2 if res is not None else 1))
    print(f'{ci:4d}->{co:4d} L={L:6d} k={k:2d} d={d}: f32 {ts["f32"]*1e3:7.1f} us ({fl/ts["f32"]/1e9:6.1f} TF)  '
          f'split {ts["split"]*1e3:7.1f} us ({fl/ts["split"]/1e9:6.1f} TF-eq, {byt/ts["split"]/1e6:6.0f} GB/s)  x{ts["f32"]/ts["split"]:.2f}')
