"""The streaming 16-channel stage + tail (v2w_stage_bf16_n16s.hip) against fp64 math and against n16_stage_kernel: parity on a few
shapes, then interleaved timing at the BASELINE configs[2] / configs[1] sizes.  (The A/B against n16_stage_kernel quoted in DESIGN.md was taken with a development switch in the dispatch, since removed: the 'old' rows of this
script now time the same kernel twice.)"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wavthruvec_pytorch_amd import hipops  # noqa: E402

dev = torch.device('cuda:0')
C, ks, d1, d2 = 16, [3, 7, 11], [1, 1, 1], [3, 3, 3]


def make(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, L, generator=g).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    wpost = torch.randn(1, C, 7, generator=g) / (C * 7) ** 0.5
    bpost = 0.1 * torch.randn(1, generator=g)
    return x, a, s, w1, b1, w2, b2, wpost, bpost


def reference(x, a, s, w1, b1, w2, b2, wpost, bpost):
    xa = (a[:, :, None] * x.float() + s[:, :, None])
    xact = F.leaky_relu(xa, 0.1).bfloat16().double()
    xres = xa.bfloat16().double()
    tot = 0
    for j, k in enumerate(ks):
        t1 = xres + F.conv1d(xact, w1[j].bfloat16().double(), b1[j].double(), padding=(k - 1) // 2)
        tact = F.leaky_relu(t1.float(), 0.1).bfloat16().double()
        tot = tot + t1 + F.conv1d(tact, w2[j].bfloat16().double(), b2[j].double(), dilation=3, padding=3 * (k - 1) // 2)
    out = tot / 3
    return torch.tanh(F.conv1d(F.leaky_relu(out, 0.01), wpost.double(), bpost.double(), padding=3))


def run(x, a, s, w1, b1, w2, b2, wpost, bpost, y):
    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b1=b1[j].to(dev),
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b2=b2[j].to(dev),
               k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(3)]
    xd, ad, sd = x.to(dev), a.to(dev), s.to(dev)
    wp, bp = wpost.permute(2, 1, 0).contiguous().to(dev), bpost.to(dev)

    def call():
        ok = hipops.resblock2_stage_split(xd, (ad, sd), br, None, slope=0.1, out_div=3.0, bf16=True, io_bf16=3, post=(wp, bp, y, 7, 0.01))
        assert ok
    return call


def parity():
    worst = 0.0
    for B, L in [(2, 1000), (3, 4100), (1, 24), (2, 64), (2, 68), (2, 216), (2, 472), (1, 948), (2, 16384), (1, 4), (3, 60)]:
        args = make(B, L, 300 + L)
        want = reference(*args)
        for off in ('', '1'):
            if off:
                os.environ['V2W_N16S_OFF'] = '1'
            else:
                os.environ.pop('V2W_N16S_OFF', None)
            y = torch.full((B, 1, L), float('nan'), device=dev)
            run(*args, y)()
            torch.cuda.synchronize()
            err = (y.cpu().double() - want).abs()
            bad = int((~torch.isfinite(y)).sum())
            print(f'B={B} L={L} {"old" if off else "new"}: max err {err.max().item():.3e} mean {err.mean().item():.3e} nonfinite {bad}', flush=True)
            if not off:
                worst = max(worst, float('inf') if bad else err.max().item())
    os.environ.pop('V2W_N16S_OFF', None)
    return worst


def timing():
    variants = [('new', {}), ('old', {'V2W_N16S_OFF': '1'})]
    for B, T in [(64, 512), (32, 256)]:
        L = T * 320
        args = make(B, L, 7)
        y = torch.empty((B, 1, L), device=dev)
        call = run(*args, y)
        res = {}
        for rnd in range(3):
            for name, env in variants:
                for k in ('V2W_N16S_OFF', 'V2W_STREAM_PRIO', 'V2W_ABL'):
                    os.environ.pop(k, None)
                os.environ.update(env)
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    call()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f'B={B} T={T}: ' + '  '.join(f'{k} {min(v):.1f}' for k, v in res.items()), flush=True)
    for k in ('V2W_N16S_OFF', 'V2W_STREAM_PRIO', 'V2W_ABL'):
        os.environ.pop(k, None)


if __name__ == '__main__':
    w = parity()
    print('worst new-kernel error', w)
    timing()
