import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes as C
from wavthruvec_pytorch_amd import hipops, _hip
dev = 'cuda'
B, cin, cout, L, k, u = 64, 512, 256, 512, 11, 5
x = torch.randn(B, cin, L, device=dev).bfloat16()
wf = torch.randn(k, cin, cout, device=dev) * 0.02
bias = torch.randn(cout, device=dev)
wps = hipops.pack_bf16_convt(wf, u)
out = torch.empty(B, cout, L * u, device=dev, dtype=torch.bfloat16)
nt = hipops.convt_bf16_stats_tiles(x, out, k, u, io_bf16=3)
part = torch.empty(nt * cout * 2, device=dev)
cfg = (C.c_int32 * 10)()
a = hipops._convt_bf16_args(x, wps, bias, out, k, u, 0.1, part, 3)
print('rc', _hip.load().v2w_convt1d_bf16_config(C.byref(a), cfg), list(cfg), 'tiles', nt)
for _ in range(3):
    hipops.convt1d_bf16(x, wps, bias, out, k=k, u=u, slope=0.1, stats_part=part, io_bf16=3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    hipops.convt1d_bf16(x, wps, bias, out, k=k, u=u, slope=0.1, stats_part=part, io_bf16=3)
e1.record(); torch.cuda.synchronize()
print(f'{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch')
