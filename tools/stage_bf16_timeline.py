#!/usr/bin/env python3
"""Where one tile of the bf16 fused stage kernel (stage_bf16_kernel, v2w_stage_bf16.hip) spends its cycles: s_memtime stamps from the
DIAGNOSTIC build (tools/stage_timeline.py build; -DV2W_TIMELINE).  Run on a GPU box:  python tools/stage_bf16_timeline.py C [IO]
(BASELINE configs[2] shapes: B = 64; C = 32 -> L = 81920, C = 16 -> L = 163840; IO = 3: bf16 activation storage)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get('V2W_TL_LIB') or os.path.join(ROOT, 'tools', 'exp', 'libv2w_timeline%s.so' % os.environ.get('V2W_TL_VARIANT', ''))
SLOTS = 32


def main(C, io):
    os.environ['V2W_LIB'] = LIB
    import numpy as np
    import torch
    from wavthruvec_pytorch_amd import _hip, hipops
    _hip.load()
    raw = ctypes.CDLL(LIB)
    stamps = hasattr(raw, 'v2w_timeline_set_stage_bf16')
    if stamps:
        raw.v2w_timeline_set_stage_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device('cuda:0')
    B, L = 64, {32: 81920, 16: 163840}[C]
    dt = torch.bfloat16 if io else torch.float32
    x = torch.randn(B, C, L, device=dev).to(dt); a = torch.rand(B, C, device=dev) + 0.5; s = torch.randn(B, C, device=dev) * 0.1
    out = torch.empty_like(x)
    br = []
    for k in (3, 7, 11):
        ws = [torch.randn(k, C, C, device=dev) / (C * k) ** 0.5 for _ in range(2)]
        br.append(dict(wps1=hipops.pack_split(ws[0], bf16=True), b1=torch.zeros(C, device=dev), wps2=hipops.pack_split(ws[1], bf16=True),
                       b2=torch.zeros(C, device=dev), k=k, dil1=1, dil2=3))
    nto = (256 - 2 * 15) & ~3
    nblk = B * ((L + nto - 1) // nto)
    buf = torch.zeros((nblk * 4 * SLOTS,), device=dev, dtype=torch.int64)
    run = lambda: hipops.resblock2_stage_split(x, (a, s), br, out, slope=0.1, out_div=3.0, bf16=True, io_bf16=io)
    assert not stamps or raw.v2w_timeline_set_stage_bf16(None, 0) == 0
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    fl = 2.0 * C * C * 42 * L * B
    print(f'stage_bf16 C={C} io={io}: {nblk} workgroups, {us:.1f} us = {fl / us / 1e6:.1f} TFLOP/s, {2 * B * C * L * (2 if io else 4) / us / 1e3:.0f} GB/s (x in + out) (stamps off)')
    if not stamps:
        return
    assert raw.v2w_timeline_set_stage_bf16(buf.data_ptr(), nblk) == 0
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print(f'  with stamps on: {e0.elapsed_time(e1) * 1e3:.1f} us')
    t = buf.cpu().numpy().reshape(nblk, 4, SLOTS).astype(np.int64)
    med = lambda v: int(np.median(v))
    print(f'  tile total {med(t[:, :, 22] - t[:, :, 0])} cycles (MFMA issue alone: {42 * (C // 16) * 2 * 32})')
    print(f'  stage x (loads + activation + LDS stores) {med(t[:, :, 1] - t[:, :, 0])}; residual re-read issue {med(t[:, :, 2] - t[:, :, 1])}; barrier {med(t[:, :, 3] - t[:, :, 2])}')
    for j, k in enumerate((3, 7, 11)):
        o = 5 * j
        prev = t[:, :, 3] if j == 0 else t[:, :, 8 + 5 * (j - 1)]
        print(f'  branch {j} (k={k}): conv1 {med(t[:, :, 4 + o] - prev)} (issue {k * (C // 16) * 2 * 32})  t1 {med(t[:, :, 5 + o] - t[:, :, 4 + o])}  '
              f'T1 stores {med(t[:, :, 6 + o] - t[:, :, 5 + o])}  barrier {med(t[:, :, 7 + o] - t[:, :, 6 + o])}  conv2 {med(t[:, :, 8 + o] - t[:, :, 7 + o])}')
    print(f'  branch sum {med(t[:, :, 20] - t[:, :, 18])}; scratch stores + barriers {med(t[:, :, 21] - t[:, :, 20])}; output {med(t[:, :, 22] - t[:, :, 21])}')


if __name__ == '__main__':
    main(int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 3)
