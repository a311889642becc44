#!/usr/bin/env python3
"""Time and (DIAGNOSTIC builds: -DV2W_TIMELINE) phase timeline of v2w_branch_convs_bf16_fwd at the BASELINE configs[2] shapes.

    python tools/res_timeline.py build [VARIANT DEFS...]   # here: tools/exp/libv2w_res<VARIANT>.so from the current sources + -D DEFS
    python tools/res_timeline.py C [VARIANT ...]            # on a GPU box: C = 128 | 64 (B = 64), every listed variant, interleaved rounds

Variants without V2W_TIMELINE are plain product builds with extra -D flags (A/B timing in one process is not possible across
libraries: each variant runs in its own child process, rounds interleaved by the parent)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SLOTS = 32
LEN = {256: 2560, 128: 10240, 64: 40960, 32: 81920, 16: 163840}


def lib_of(variant):
    return os.path.join(ROOT, 'tools', 'exp', f'libv2w_res{variant}.so')


def build(variant, defs):
    from wavthruvec_pytorch_amd import build as b
    os.makedirs(os.path.join(ROOT, 'tools', 'exp'), exist_ok=True)
    cmd = [b.find_hipcc()] + b.FLAGS + ['-w'] + [f'-D{d}' for d in defs] + ['-shared', '-o', lib_of(variant)] + [os.path.join(b.CSRC, s) for s in b.SOURCES]
    subprocess.run(cmd, check=True)
    print(lib_of(variant))


def child(C, variant):
    os.environ['V2W_LIB'] = lib_of(variant)
    import numpy as np
    import torch
    from wavthruvec_pytorch_amd import _hip, hipops
    _hip.load()
    raw = ctypes.CDLL(lib_of(variant))
    stamps = hasattr(raw, 'v2w_timeline_set_res')
    if stamps:
        raw.v2w_timeline_set_res.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert raw.v2w_timeline_set_res(None, 0) == 0
    dev = torch.device('cuda:0')
    B, L = 64, LEN[C]
    ks, d0, d1 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    x = torch.randn(B, C, L, device=dev).bfloat16()
    a = torch.rand(B, C, device=dev) + 0.5
    s = torch.randn(B, C, device=dev) * 0.1
    wps = [hipops.pack_split(torch.randn(k, C, C, device=dev) / (C * k) ** 0.5, bf16=True)[0] for k in ks]
    bs = [torch.zeros(C, device=dev) for _ in ks]
    t1 = [torch.empty_like(x) for _ in ks]
    out = torch.empty_like(x)
    runs = {0: lambda: hipops.branch_convs_bf16(0, [x], (a, s), wps, bs, t1, ks, d0, slope=0.1),
            1: lambda: hipops.branch_convs_bf16(1, t1, None, wps, bs, [out], ks, d1, slope=0.1, out_div=3.0)}
    fl = 2.0 * C * C * sum(ks) * L * B
    res = {}
    for mode, run in runs.items():
        for _ in range(3):
            run()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    print(f'[{variant or "base"}] C={C}: mode0 {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF)   mode1 {res[1]:7.1f} us ({fl / res[1] / 1e6:6.1f} TF)', flush=True)
    if not stamps:
        return
    nt = 256
    nblk = ((B * ((L + nt - 1) // nt) + 7) // 8 * 8) * (1 if C <= 128 else 2)
    buf = torch.zeros((nblk * 4 * SLOTS,), device=dev, dtype=torch.int64)
    med = lambda v: int(np.median(v))
    for mode, run in runs.items():
        buf.zero_()
        assert raw.v2w_timeline_set_res(buf.data_ptr(), nblk) == 0
        run(); torch.cuda.synchronize()
        assert raw.v2w_timeline_set_res(None, 0) == 0
        t = buf.cpu().numpy().reshape(nblk, 4, SLOTS).astype(np.int64)
        t = t[t[:, 0, 0] != 0]
        d = lambda i, j: med(t[:, :, i] - t[:, :, j])
        print(f'  mode {mode}: tile total {d(20, 0)} cycles over {t.shape[0]} tiles')
        if mode == 0:
            print(f'    staging (loads + commits) {d(1, 0)}  barrier {d(2, 1)}')
            for j, k in enumerate(ks):
                print(f'    branch {j} (k={k}): init {d(3 + 3 * j, 2 + 3 * j)}  conv {d(4 + 3 * j, 3 + 3 * j)} (issue alone {k * (C // 32) * 2 * 8 * 32})  store {d(5 + 3 * j, 4 + 3 * j)}')
        else:
            for j, k in enumerate(ks):
                pre = f'wait {d(1 + 5 * j, 5 * j)}  ' if j else ''
                print(f'    branch {j} (k={k}): {pre}staging {d(2 + 5 * j, 1 + 5 * j)}  barrier {d(3 + 5 * j, 2 + 5 * j)}  residual {d(4 + 5 * j, 3 + 5 * j)}  '
                      f'conv {d(5 + 5 * j, 4 + 5 * j)} (issue alone {k * (C // 32) * 2 * 8 * 32})')
            print(f'    store {d(20, 15)}')
        clk = (t[:, 0, 20] - t[:, 0, 0]).astype(float) / np.maximum(1, (t[:, 0, SLOTS - 3] * 0 + 1))
        span = t[:, :, 20].max() - t[:, :, 0].min()
        print(f'    kernel span {span} cycles')


def child_stage(C, variant):
    """The fused wide stage kernel (v2w_stage_bf16_wide.hip) at the configs[2] shape of stage C."""
    os.environ['V2W_LIB'] = lib_of(variant)
    import numpy as np
    import torch
    from wavthruvec_pytorch_amd import _hip, hipops
    _hip.load()
    raw = ctypes.CDLL(lib_of(variant))
    n16 = C == 16 and hasattr(raw, 'v2w_timeline_set_n16') and not os.environ.get('V2W_TL_WIDE16')
    n32 = C == 32 and hasattr(raw, 'v2w_timeline_set_n32') and not os.environ.get('V2W_TL_WIDE32')
    setter = getattr(raw, 'v2w_timeline_set_n16' if n16 else ('v2w_timeline_set_n32' if n32 else 'v2w_timeline_set_wide'), None)
    stamps = setter is not None
    if stamps:
        setter.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert setter(None, 0) == 0
    dev = torch.device('cuda:0')
    B, L = 64, LEN[C]
    ks = [3, 7, 11]
    x = torch.randn(B, C, L, device=dev).bfloat16()
    a = torch.rand(B, C, device=dev) + 0.5
    s = torch.randn(B, C, device=dev) * 0.1
    br = [dict(wps1=hipops.pack_split(torch.randn(k, C, C, device=dev) / (C * k) ** 0.5, bf16=True), b1=torch.zeros(C, device=dev),
               wps2=hipops.pack_split(torch.randn(k, C, C, device=dev) / (C * k) ** 0.5, bf16=True), b2=torch.zeros(C, device=dev),
               k=k, dil1=1, dil2=3) for k in ks]
    out = torch.empty_like(x)
    fused_up = os.environ.get('V2W_TL_UP') and C >= 32           # the stage with the next upsampler behind it (stride 4 after C >= 128, else 2)
    if fused_up:
        u = 4 if C >= 128 else 2
        wpu = hipops.pack_bf16_convt(torch.randn(2 * u, C, C // 2, device=dev) / (C * 2) ** 0.5, u)
        nt = hipops.resblock2_stage_up_tiles(B, C, L, ks, [1, 1, 1], [3, 3, 3], slope=0.1, up_k=2 * u, up_u=u, up_slope=0.1)
        assert nt > 0
        uout = torch.empty((B, C // 2, L * u), device=dev, dtype=torch.bfloat16)
        part = torch.empty((nt * (C // 2) * 2,), device=dev)
        run = lambda: hipops.resblock2_stage_split(x, (a, s), br, None, slope=0.1, out_div=3.0, bf16=True, io_bf16=3,
                                                   up=(wpu, torch.zeros(C // 2, device=dev), uout, part, 2 * u, u, 0.1))
    else:
        run = lambda: hipops.resblock2_stage_split(x, (a, s), br, out, slope=0.1, out_div=3.0, bf16=True, io_bf16=3)
    assert run()
    fl = 2 * 2.0 * C * C * sum(ks) * L * B
    for _ in range(3):
        run()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    print(f'[{variant or "base"}] stage C={C}: {us:7.1f} us ({fl / us / 1e6:6.1f} TF useful)', flush=True)
    if not stamps:
        return
    W = {128: 256, 64: 256, 256: 128, 32: 256, 16: 256}[C]
    nto = (W - 30) & ~3
    nblk = B * ((L + nto - 1) // nto)
    buf = torch.zeros((nblk * 4 * SLOTS,), device=dev, dtype=torch.int64)
    assert setter(buf.data_ptr(), nblk) == 0
    run(); torch.cuda.synchronize()
    assert setter(None, 0) == 0
    t = buf.cpu().numpy().reshape(nblk, 4, SLOTS).astype(np.int64)
    t = t[t[:, 0, 0] != 0]
    med = lambda v: f'{int(np.median(v))}/{int(np.mean(v))}'
    tw = t.reshape(-1, SLOTS)
    tw = tw[tw[:, 0] != 0]                                  # (workgroups of two waves leave the slots of waves 2-3 empty)
    d = lambda i, j: med(tw[:, i] - tw[:, j])
    if n32:     # conv1 waves of the role-specialised kernel, the last full iteration of each workgroup is overwritten by the drain iteration: read with care
        print(f'  (median/mean cycles, conv1 waves, last iteration of {t.shape[0]} workgroups)  iteration {d(13, 0)}')
        print(f'    slot 0: sum -> SF + commit {d(1, 0)}  wait {d(2, 1)}')
        print(f'    slot 1: conv1_0 {d(3, 2)}  epilogue {d(4, 3)}  wait {d(5, 4)}')
        print(f'    slot 2: conv1_1 {d(6, 5)}  epilogue {d(7, 6)}  wait {d(8, 7)}')
        print(f'    slot 3: conv1_2 {d(9, 8)}  issue + epilogue {d(10, 9)}  wait {d(13, 10)}')
        return
    if n16:     # persistent workgroups: the stamps of each workgroup's LAST tile
        print(f'  (median/mean cycles, last tile of {t.shape[0]} workgroups)  tile total {d(20, 0)}; staging {d(1, 0)}  barrier {d(2, 1)}')
        for j, k in enumerate(ks):
            print(f'    branch {j} (k={k}): conv1 {d(3 + 5 * j, 2 if j == 0 else 7 + 5 * (j - 1))} (issue alone {((k + 1) // 2) * 8 * 16})  barrier {d(4 + 5 * j, 3 + 5 * j)}  '
                  f't1 {d(5 + 5 * j, 4 + 5 * j)}  barrier {d(6 + 5 * j, 5 + 5 * j)}  conv2 {d(7 + 5 * j, 6 + 5 * j)}')
        print(f'    barrier {d(18, 17)}  scratch {d(19, 18)}  store {d(20, 19)}')
        return
    nch = max(1, C // 32)
    mi, ni = 2, 2
    print('  (median/mean cycles)')
    print(f'  tile total {d(28, 0)} cycles over {t.shape[0]} tiles (waves 0-3 of 8 stamped); staging {d(1, 0)}  barrier {d(2, 1)}')
    for j, k in enumerate(ks):
        iss = k * nch * 2 * mi * ni * 32
        print(f'    branch {j} (k={k}): init {d(3 + 6 * j, 2 if j == 0 else 8 + 6 * (j - 1))}  conv1 {d(4 + 6 * j, 3 + 6 * j)} (issue alone {iss})  barrier {d(5 + 6 * j, 4 + 6 * j)}  '
              f't1 {d(6 + 6 * j, 5 + 6 * j)}  barrier {d(7 + 6 * j, 6 + 6 * j)}  conv2 {d(8 + 6 * j, 7 + 6 * j)}')
    if fused_up:
        print(f'    barrier {d(27, 20)}  z tile + barrier {d(21, 27)}  upsampler conv {d(22, 21)}  epilogue + stats {d(23, 22)}   tile total {d(23, 0)}')
        return
    print(f'    barrier {d(27, 20)}  store {d(28, 27)}')
    clk = (t[:, 0, 28] - t[:, 0, 0])
    print(f'    clock: kernel {us:.1f} us; tiles per CU {t.shape[0] / 256:.2f}; sum of tile cycles per CU / kernel time = {np.sum(clk) / 256 / us / 1e3:.2f} GHz-equivalent')


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2] if len(sys.argv) > 2 else '', sys.argv[3:])
    elif sys.argv[1] == 'child':
        child(int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else '')
    elif sys.argv[1] == 'child_stage':
        child_stage(int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else '')
    elif sys.argv[1] == 'stage':
        for rnd in range(2):
            for v in (sys.argv[3:] or ['']):
                subprocess.run([sys.executable, os.path.abspath(__file__), 'child_stage', sys.argv[2], v])
    else:
        C = int(sys.argv[1])
        variants = sys.argv[2:] or ['']
        for rnd in range(2):
            for v in variants:
                subprocess.run([sys.executable, os.path.abspath(__file__), 'child', str(C), v])
