#!/usr/bin/env python3
"""Where one tile of the fused narrow-stage kernel spends its cycles: s_memtime stamps from a DIAGNOSTIC build.

    python tools/stage_timeline.py build      # here (no GPU): hipcc -DV2W_TIMELINE -> tools/exp/libv2w_timeline.so (travels)
    V2W_TL_VARIANT=_nolrelu V2W_TL_DEFS=V2W_TL_NOLRELU python tools/stage_timeline.py build      # a what-if variant next to it
    V2W_TL_LDSPAD=40000 python tools/stage_timeline.py 32    # pad the dynamic LDS: one workgroup per CU = one wave per SIMD
    python tools/stage_timeline.py [C]        # on a GPU box: run resblock2_stage at the cfg2 shape of stage C (32 | 16), print

The product library is never instrumented: V2W_STAMP compiles to nothing without -DV2W_TIMELINE (csrc/v2w_common.h).
Output: per phase the median / p90 cycles over all waves, how many workgroups share a CU at the same time, and how far apart
(in phase) the co-resident workgroups of one CU run - lockstep neighbours cannot hide each other's non-MFMA phases.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'exp', 'libv2w_timeline%s.so' % os.environ.get('V2W_TL_VARIANT', ''))
SLOTS = 32


def build():
    from wavthruvec_pytorch_amd import build as b
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [b.find_hipcc()] + b.FLAGS + ['-DV2W_TIMELINE', '-w'] + [f'-D{d}' for d in os.environ.get('V2W_TL_DEFS', '').split()] + ['-shared', '-o', LIB] + [os.path.join(b.CSRC, s) for s in b.SOURCES]
    subprocess.run(cmd, check=True)
    print(LIB)


def main(C):
    os.environ['V2W_LIB'] = LIB
    import numpy as np
    import torch
    from wavthruvec_pytorch_amd import _hip, hipops
    lib = _hip.load()
    raw = ctypes.CDLL(LIB)
    raw.v2w_timeline_set.argtypes = [ctypes.c_void_p, ctypes.c_int]
    flags = os.environ.get('V2W_TL_VARIANT', '')
    dev = torch.device('cuda:0')
    B, L = 32, {32: 40960, 16: 81920}[C]
    x = torch.randn(B, C, L, device=dev); a = torch.ones(B, C, device=dev); s = torch.zeros(B, C, device=dev)
    out = torch.empty_like(x)
    br = []
    for k in (3, 7, 11):
        ws = [torch.randn(k, C, C, device=dev) / (C * k) ** 0.5 for _ in range(2)]
        bs = [torch.zeros(C, device=dev) for _ in range(2)]
        br.append(dict(wp1=hipops.pack_mfma(ws[0]), b1=bs[0], wp2=hipops.pack_mfma(ws[1]), b2=bs[1], k=k, dil1=1, dil2=3))
    nto = 224
    nblk = B * ((L + nto - 1) // nto)
    buf = torch.zeros((nblk * 4 * SLOTS,), device=dev, dtype=torch.int64)
    run = lambda: hipops.resblock2_stage(x, (a, s), br, out, slope=0.1, out_div=3.0)
    assert raw.v2w_timeline_set(None, 0) == 0
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print(f'flags={flags} ldspad={os.environ.get("V2W_TL_LDSPAD")} C={C} L={L}: {nblk} workgroups, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (stamps off)')
    assert raw.v2w_timeline_set(buf.data_ptr(), nblk) == 0
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print(f'  with stamps on: {e0.elapsed_time(e1) * 1e3:.1f} us')
    t = buf.cpu().numpy().reshape(nblk, 4, SLOTS).astype(np.int64)
    st = t[:, :, :16]
    names = ['stage x (loads+LDS writes)', 'barrier'] + sum([[f'b{j} conv1 (k={k})', f'b{j} epi1+T1+barriers', f'b{j} conv2', f'b{j} epi2']
                                                             for j, k in enumerate((3, 7, 11))], []) + ['store']
    d = np.diff(st, axis=2)                      # (nblk, 4, 15)
    tot = st[:, :, 15] - st[:, :, 0]
    print(f'  tile total: median {np.median(tot):.0f} cycles, p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}')
    mf = {32: 64, 16: 32}[C] * 2 * (16 if C == 32 else 4)     # cycles of MFMA issue per tap and wave: NI x QT x cycles
    for i, n in enumerate(names):
        extra = ''
        if 'conv' in n:
            k = (3, 7, 11)[int(n[1])]
            extra = f'   (MFMA issue alone: {k * mf * (2 if C == 32 else 4) // 2} cycles)'
        print(f'  {n:30s} median {np.median(d[:, :, i]):8.0f}  p90 {np.percentile(d[:, :, i], 90):8.0f}{extra}')
    if C == 32:
        tp = t[:, :, 16:30]
        print('  per-tap cycles inside b0 conv1 (k=3):', [int(np.median(tp[:, :, 0] - st[:, :, 2]))] + [int(np.median(tp[:, :, i] - tp[:, :, i - 1])) for i in (1, 2)])
        print('  per-tap cycles inside b2 conv1 (k=11):', [int(np.median(tp[:, :, 3] - st[:, :, 10]))] + [int(np.median(tp[:, :, i] - tp[:, :, i - 1])) for i in range(4, 14)])
    # co-residency: HW_ID bits (gfx9): wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13 ; XCC_ID 3:0
    hw, xcc = t[:, 0, SLOTS - 1], t[:, 0, SLOTS - 2] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    real0 = t[:, 0, SLOTS - 3]
    clk = (st[:, 0, 15] - st[:, 0, 0]).sum() / max(1, 1)   # cycles
    print(f'  distinct CUs seen: {len(set(cu.tolist()))}')
    # phase offset between consecutive workgroups on the same CU, in units of the tile duration
    offs = []
    for c in set(cu.tolist()):
        idx = np.where(cu == c)[0]
        starts = np.sort(st[idx, 0, 0])
        if len(starts) > 2:
            offs += list(np.diff(starts) / np.median(tot))
    offs = np.array(offs)
    hist, edges = np.histogram(np.clip(offs, 0, 1.5), bins=15, range=(0, 1.5))
    print('  start-to-start offset of successive workgroups on one CU (fraction of a tile time): histogram 0..1.5 in 0.1 bins')
    print('   ', hist.tolist())


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        main(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
