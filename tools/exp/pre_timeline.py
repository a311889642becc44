"""Where a tile of conv_pre (conv_bf16_kernel, fp32 latents in, bf16 out) spends its cycles at the cfg3 / cfg2 shapes: s_memtime stamps of the
DIAGNOSTIC build (python tools/stage_timeline.py build; -DV2W_TIMELINE).  Per chunk: MFMA phase, commit, barrier.  tools/exp."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'exp', 'libv2w_timeline.so')
SLOTS = 32
os.environ['V2W_LIB'] = LIB
import numpy as np
import torch
from wavthruvec_pytorch_amd import _hip, hipops
_hip.load()
raw = ctypes.CDLL(LIB)
raw.v2w_timeline_set_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device('cuda:0')
for B, T in ((64, 512), (32, 256)):
    cin = 768
    x = torch.randn(B, cin, T, device=dev)
    v = torch.randn(512, cin, 7, device=dev) * 0.02
    frag, sc = torch.empty(hipops.split_halves(7, cin, 512) + 1024, device=dev, dtype=torch.float16), torch.empty(4, device=dev)
    hipops.SplitPlan([(v, None, frag, sc)], dev, bf16=True).run()
    out = torch.empty(B, 512, T, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(512, device=dev)
    name = hipops.conv_bf16_config(B, 1, cin, 512, T, 7, 1, 1, io_bf16=2)
    cfg = [int(s) for s in name[name.index('<') + 1:].split(',')[:5]]
    mi, ni, wm, wn = cfg[:4]
    mt, nt = 32 * mi * wm, 32 * ni * wn
    ck = int(name.split(',')[8])
    nch = cin // ck
    run = lambda: hipops.conv1d(x, None, bias, out, k=7, dil=1, slope=1.0, algo=hipops.ALGO_BF16, wps=(frag, sc), io_bf16=2)
    nblk = ((B * ((T + nt - 1) // nt) + 7) // 8 * 8) * (512 // mt)
    buf = torch.zeros((nblk * 4 * SLOTS,), device=dev, dtype=torch.int64)
    assert raw.v2w_timeline_set_bf16(None, 0) == 0
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print(f'{name}: B={B} T={T}: {nblk} workgroups of {mt} x {nt}, {nch} chunks of {ck} channels; {e0.elapsed_time(e1) * 1e3:.1f} us (stamps off)')
    assert raw.v2w_timeline_set_bf16(buf.data_ptr(), nblk) == 0
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    assert raw.v2w_timeline_set_bf16(None, 0) == 0
    print(f'  with stamps on: {e0.elapsed_time(e1) * 1e3:.1f} us')
    t = buf.cpu().numpy().reshape(nblk, 4, SLOTS).astype(np.int64)
    t = t[t[:, 0, 0] != 0]
    med = lambda a: int(np.median(a))
    print(f'  tile total {med(t[:, :, 27] - t[:, :, 0])} cycles of the 100 MHz counter x 24 = GPU cycles?; prologue {med(t[:, :, 1] - t[:, :, 0])}; last stamped chunk end -> tile end {med(t[:, :, 27] - t[:, :, 5 + 4 * 5])}')
    ideal = 7 * (ck // 16) * mi * ni * 32
    for c in range(6):
        line = f'  chunk {c}: prefetch issue {med(t[:, :, 2 + 4 * c] - (t[:, :, 1] if c == 0 else t[:, :, 5 + 4 * (c - 1)])):6d}  MFMA phase {med(t[:, :, 3 + 4 * c] - t[:, :, 2 + 4 * c]):7d} (MFMA issue alone {ideal} GPU cycles)'
        line += f'  commit {med(t[:, :, 4 + 4 * c] - t[:, :, 3 + 4 * c]):6d}  barrier {med(t[:, :, 5 + 4 * c] - t[:, :, 4 + 4 * c]):6d}'
        print(line)
    print(f'  kernel span {t[:, :, 27].max() - t[:, :, 0].min()} counter ticks')
