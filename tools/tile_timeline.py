#!/usr/bin/env python3
"""Where one tile of the f32 MFMA conv kernel (conv_tile_kernel) spends its cycles: s_memtime stamps from the DIAGNOSTIC build
(tools/stage_timeline.py build; -DV2W_TIMELINE).  Run on a GPU box:

    python tools/tile_timeline.py C K DIL [NPROB]      # one generator conv layer shape at cfg2: C_in = C_out = C, B = 32

Per chunk: MFMA phase (with the next chunk's global loads in flight), commit (affine + leaky_relu + LDS stores), barrier.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get('V2W_TL_LIB') or os.path.join(ROOT, 'tools', 'exp', 'libv2w_timeline%s.so' % os.environ.get('V2W_TL_VARIANT', ''))
SLOTS = 32
LEN = {256: 1280, 128: 5120, 64: 20480, 32: 40960, 16: 81920}


def main(C, K, dil, nprob):
    os.environ['V2W_LIB'] = LIB
    import numpy as np
    import torch
    from wavthruvec_pytorch_amd import _hip, hipops
    _hip.load()
    raw = ctypes.CDLL(LIB)
    stamps = hasattr(raw, 'v2w_timeline_set_tile')       # (a product build of another revision can be timed too: no stamps)
    if stamps:
        raw.v2w_timeline_set_tile.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device('cuda:0')
    B, L = 32, LEN[C]
    x = torch.randn(B, C, L, device=dev); a = torch.rand(B, C, device=dev) + 0.5; s = torch.randn(B, C, device=dev) * 0.1
    probs = []
    for q in range(nprob):
        wf = torch.randn(K, C, C, device=dev) / (C * K) ** 0.5
        probs.append((x, wf, torch.zeros(C, device=dev), torch.empty_like(x),
                      dict(k=K, dil=dil, slope=0.1, in_affine=(a, s), res=x, res_affine=(a, s), wp=hipops.pack_mfma(wf))))
    run = (lambda: hipops.conv1d_multi(probs)) if nprob > 1 else (lambda: hipops.conv1d(*probs[0][:4], **probs[0][4]))
    cfg = hipops.conv_tile_config(B * nprob, C, C, L, K, dil)
    mt = int(cfg.split('<')[1].split(',')[0]) * int(cfg.split(',')[2]) * int(cfg.split(',')[4])
    nt = int(cfg.split('<')[1].split(',')[0]) * int(cfg.split(',')[3]) * int(cfg.split(',')[5])
    nblk = nprob * ((B * ((L + nt - 1) // nt) + 7) // 8 * 8) * (C // mt)
    buf = torch.zeros((nblk * 4 * SLOTS,), device=dev, dtype=torch.int64)
    assert not stamps or raw.v2w_timeline_set_tile(None, 0) == 0
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    fl = 2.0 * C * C * K * L * B * nprob
    print(f'{cfg}  C={C} K={K} dil={dil} x{nprob}: {nblk} workgroups, {us:.1f} us = {fl / us / 1e6:.1f} TFLOP/s (stamps off)')
    if not stamps:
        return
    assert raw.v2w_timeline_set_tile(buf.data_ptr(), nblk) == 0
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print(f'  with stamps on: {e0.elapsed_time(e1) * 1e3:.1f} us')
    t = buf.cpu().numpy().reshape(nblk, 4, SLOTS).astype(np.int64)
    t = t[t[:, 0, 0] != 0]                      # (tiles past the end return before the first stamp)
    nch = C // 32
    med = lambda v: int(np.median(v))
    print(f'  tile total {med(t[:, :, 27] - t[:, :, 0])} cycles; prologue (tables + chunk 0 staging + barrier) {med(t[:, :, 1] - t[:, :, 0])}; '
          f'epilogue {med(t[:, :, 27] - t[:, :, 26])}')
    mi, ni = int(cfg.split(',')[2]), int(cfg.split(',')[3])
    ideal = K * 4 * 4 * mi * ni * 64
    for c in range(min(nch, 6)):
        line = f'  chunk {c}: prefetch issue {med(t[:, :, 2 + 4 * c] - (t[:, :, 1] if c == 0 else t[:, :, 5 + 4 * (c - 1)])):6d}  MFMA phase {med(t[:, :, 3 + 4 * c] - t[:, :, 2 + 4 * c]):7d} (issue alone {ideal})'
        if c + 1 < nch:
            line += f'  commit {med(t[:, :, 4 + 4 * c] - t[:, :, 3 + 4 * c]):6d}  barrier {med(t[:, :, 5 + 4 * c] - t[:, :, 4 + 4 * c]):6d}'
        print(line)


if __name__ == '__main__':
    C, K, dil = (int(v) for v in sys.argv[1:4])
    main(C, K, dil, int(sys.argv[4]) if len(sys.argv) > 4 else 1)
