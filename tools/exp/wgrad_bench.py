#!/usr/bin/env python3
"""Isolated time of every Conv1d weight-gradient launch of a cfg2 training step (one queue, nothing beside it): the exact fp32 kernel,
the bf16-operand kernel on fp32 tensors and on bf16 tensors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from wavthruvec_pytorch_amd import hipops  # noqa: E402

B = 32
dev = torch.device('cuda:0')
def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100


tot = [0.0, 0.0, 0.0]
for C, L in ((256, 1280), (128, 5120), (64, 20480), (32, 40960), (16, 81920)):
    x = torch.randn(B, C, L, device=dev)
    dy = torch.randn(B, C, L, device=dev)
    xb, dyb = x.bfloat16(), dy.bfloat16()
    a = torch.rand(B * C, device=dev) + 0.5
    s = torch.randn(B * C, device=dev)
    st = [0.0, 0.0, 0.0]
    for k in (3, 7, 11):
        for d in (1, 3):
            aff = (a, s) if d == 1 else None
            us = [timed(lambda: hipops.wgrad(x, dy, k=k, dil=d, slope=0.1, x_affine=aff)),
                  timed(lambda: hipops.wgrad_bf16(x, dy, k=k, dil=d, slope=0.1, x_affine=aff)),
                  timed(lambda: hipops.wgrad_bf16(xb, dyb, k=k, dil=d, slope=0.1, x_affine=aff))]
            gf = 2.0 * C * C * k * L * B / 1e9
            mb = 2 * x.numel() * 4 / 1e6
            print(f'C={C:3d} L={L:5d} k={k:2d} d={d}: f32 {us[0]:7.1f} us {gf / us[0] * 1e3 / 1e3:6.1f} TF | bf16 ops, f32 tensors {us[1]:7.1f} us {gf / us[1]:6.1f} TF '
                  f'{mb / us[1]:5.2f} TB/s | bf16 tensors {us[2]:7.1f} us {gf / us[2]:6.1f} TF {mb / 2 / us[2]:5.2f} TB/s')
            for i in range(3):
                st[i] += us[i]
    print(f'  stage C={C}: {st[0] / 1e3:.3f} / {st[1] / 1e3:.3f} / {st[2] / 1e3:.3f} ms')
    for i in range(3):
        tot[i] += st[i]
print(f'all ResBlock2 weight gradients: {tot[0] / 1e3:.3f} / {tot[1] / 1e3:.3f} / {tot[2] / 1e3:.3f} ms')
