import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from wavthruvec_pytorch_amd import hipops, _hip
dev = torch.device('cuda:0')
B, ci, co, L, k, d = int(os.environ.get('TL_B', '32')), 256, 256, 1024, 11, 1
x = torch.randn(B, ci, L, device=dev); wf = torch.randn(k, ci, co, device=dev) / (ci * k) ** 0.5
out = torch.empty(B, co, L, device=dev); wps = hipops.pack_split(wf)
for _ in range(3): hipops.conv1d(x, None, None, out, k=k, dil=d, slope=0.1, res=x, algo=hipops.ALGO_SPLIT, wps=wps)
torch.cuda.synchronize()
lib = _hip.load(); buf = np.zeros(8192, dtype=np.uint64)
lib._handle if False else None
f = ctypes.CDLL(_hip.lib_path()).v2w_debug_read; f.argtypes = [ctypes.c_void_p, ctypes.c_int]
print('rc', f(buf.ctypes.data, buf.nbytes))
t = buf[:2048].reshape(4, 64, 8).astype(np.int64)
names = ['start', 'dma+sig issued', 'lds reads done', 'mfma issued', 'commit done', 'vmcnt wait done', 'barrier done']
rt = t[0, 40, 7] - t[0, 10, 7]; sc = t[0, 40, 0] - t[0, 10, 0]
print(f'30 stages: {sc} shader cycles, {rt} realtime ticks (100 MHz) -> shader clock {sc / (rt / 100.0):.0f} MHz, {sc / 30:.0f} cycles/stage')
for w in range(1):
    print('wave', w)
    for st in range(20, 28):
        r = t[w, st]
        print(f'  st {st:2d}: ' + '  '.join(f'{names[i + 1]}: +{r[i + 1] - r[i]:5d}' for i in range(6)) + f'   | stage total {t[w, st + 1, 0] - r[0]:5d}')
