#!/usr/bin/env python3
"""Where one tile of the bf16 conv kernel (conv_bf16_kernel, v2w_conv_bf16.hip) spends its cycles: s_memtime stamps from the
DIAGNOSTIC build (tools/stage_timeline.py build; -DV2W_TIMELINE).  Run on a GPU box:

    python tools/bf16_timeline.py C K DIL [IO]      # a residual conv of BASELINE configs[2]: C_in = C_out = C, B = 64; IO = 3: bf16 storage

Per chunk: MFMA phase (with the next chunk's global loads in flight), commit (affine + leaky_relu + bf16 + LDS stores), barrier.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get('V2W_TL_LIB') or os.path.join(ROOT, 'tools', 'exp', 'libv2w_timeline%s.so' % os.environ.get('V2W_TL_VARIANT', ''))
SLOTS = 32
LEN = {256: 2560, 128: 10240, 64: 40960}


def main(C, K, dil, io):
    os.environ['V2W_LIB'] = LIB
    import numpy as np
    import torch
    from wavthruvec_pytorch_amd import _hip, hipops
    _hip.load()
    raw = ctypes.CDLL(LIB)
    stamps = hasattr(raw, 'v2w_timeline_set_bf16')
    if stamps:
        raw.v2w_timeline_set_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device('cuda:0')
    B, L = 64, LEN[C]
    dt = torch.bfloat16 if io else torch.float32
    x = torch.randn(B, C, L, device=dev).to(dt); a = torch.rand(B, C, device=dev) + 0.5; s = torch.randn(B, C, device=dev) * 0.1
    wf = torch.randn(K, C, C, device=dev) / (C * K) ** 0.5
    out = torch.empty_like(x)
    kw = dict(k=K, dil=dil, slope=0.1, in_affine=(a, s), res=x, res_affine=(a, s), algo=hipops.ALGO_BF16, wps=hipops.pack_split(wf, bf16=True),
              io_bf16=io)
    run = lambda: hipops.conv1d(x, wf, torch.zeros(C, device=dev), out, **kw)
    big = C % 128 == 0 and B * ((L + 255) // 256) * (C // 64) >= 1024
    mt, nt, mi, ni = (128, 256, 2, 4) if big else (64, 256, 1, 4)
    nblk = ((B * ((L + nt - 1) // nt) + 7) // 8 * 8) * (C // mt)
    buf = torch.zeros((nblk * 4 * SLOTS,), device=dev, dtype=torch.int64)
    assert not stamps or raw.v2w_timeline_set_bf16(None, 0) == 0
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    fl = 2.0 * C * C * K * L * B
    by = (2 + (1 if True else 0)) * B * C * L * (2 if io else 4)
    print(f'{mt}x{nt}  C={C} K={K} dil={dil} io={io}: {nblk} workgroups, {us:.1f} us = {fl / us / 1e6:.1f} TFLOP/s, {by / us / 1e3:.0f} GB/s (in + res + out) (stamps off)')
    if not stamps:
        return
    assert raw.v2w_timeline_set_bf16(buf.data_ptr(), nblk) == 0
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print(f'  with stamps on: {e0.elapsed_time(e1) * 1e3:.1f} us')
    t = buf.cpu().numpy().reshape(nblk, 4, SLOTS).astype(np.int64)
    t = t[t[:, 0, 0] != 0]
    nch = C // 32
    med = lambda v: int(np.median(v))
    print(f'  tile total {med(t[:, :, 27] - t[:, :, 0])} cycles; prologue (tables + chunk 0 staging + barrier) {med(t[:, :, 1] - t[:, :, 0])}; '
          f'epilogue {med(t[:, :, 27] - t[:, :, 26])}')
    ideal = K * 2 * mi * ni * 32
    for c in range(min(nch, 6)):
        line = f'  chunk {c}: prefetch issue {med(t[:, :, 2 + 4 * c] - (t[:, :, 1] if c == 0 else t[:, :, 5 + 4 * (c - 1)])):6d}  MFMA phase {med(t[:, :, 3 + 4 * c] - t[:, :, 2 + 4 * c]):7d} (issue alone {ideal})'
        if c + 1 < nch:
            line += f'  commit {med(t[:, :, 4 + 4 * c] - t[:, :, 3 + 4 * c]):6d}  barrier {med(t[:, :, 5 + 4 * c] - t[:, :, 4 + 4 * c]):6d}'
        print(line)
    # how many workgroups ran per CU at once: distinct (xcc, cu) pairs vs workgroups alive at the median start time
    span = t[:, :, 27].max() - t[:, :, 0].min()
    print(f'  kernel span {span} cycles (100 MHz-corrected clock not applied); waves/tile 4; tiles {t.shape[0]}')


if __name__ == '__main__':
    C, K, dil = (int(v) for v in sys.argv[1:4])
    main(C, K, dil, int(sys.argv[4]) if len(sys.argv) > 4 else 3)
