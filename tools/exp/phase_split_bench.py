import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavthruvec_pytorch_amd import _hip
lib = _hip.load()
dev = 'cuda'
def pitch(n): return (n + 3) // 4 * 4
cases = [('msd L1', 32, 128, 32, 81920, 1, 2), ('msd L3', 32, 256, 16, 20480, 1, 4), ('msd L4', 32, 512, 32, 5120, 1, 4),
         ('mpd13 L1', 32, 32, 32, 2101, 13, 3), ('mpd19 L2', 32, 128, 128, 480, 19, 3), ('mpd17 L3', 32, 512, 512, 179, 17, 3)]
for name, B, C, Cg, L, inner, s in cases:
    U = -(-L // s)
    ip, op = pitch(L * inner), pitch(U * inner)
    x = torch.randn(B, C, ip, device=dev)
    out = torch.empty(B, s * C, op, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _hip.check(lib.v2w_phase_split(x.data_ptr(), out.data_ptr(), B, C, Cg, L, inner, s, ip, op, st), 'ps')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.v2w_phase_split(x.data_ptr(), out.data_ptr(), B, C, Cg, L, inner, s, ip, op, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = (x.numel() + out.numel()) * 4 / 1e9
    print(f'{name:10s} {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  checksum {out.double().sum().item():.6e}')
