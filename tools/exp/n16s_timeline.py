"""Where one steady-state step of the streaming 16-channel kernel spends its cycles (s_memtime stamps, -DV2W_TIMELINE build:
python tools/stage_timeline.py build).  Median over the waves of the launch."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'exp', 'libv2w_timeline.so')
os.environ['V2W_LIB'] = LIB
import numpy as np  # noqa: E402
import torch  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import n16s_check as n  # noqa: E402
from wavthruvec_pytorch_amd import _hip  # noqa: E402

_hip.load()
raw = ctypes.CDLL(LIB)
raw.v2w_timeline_set_n16s.argtypes = [ctypes.c_void_p, ctypes.c_int]
B, T = 64, 512
L = T * 320
y = torch.empty((B, 1, L), device=n.dev)
call = n.run(*n.make(B, L, 7), y)
nblk = 1792
buf = torch.zeros((nblk * 4 * 32,), device=n.dev, dtype=torch.int64)
assert raw.v2w_timeline_set_n16s(None, 0) == 0
for _ in range(3):
    call()
assert raw.v2w_timeline_set_n16s(buf.data_ptr(), nblk) == 0
call()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nblk, 4, 32)[:, 0, :].astype(np.int64)
ok = t[:, 6] > t[:, 0]
t = t[ok]
med = lambda v: int(np.median(v))
names = ['wait for the last step\'s LDS writes', 'operand reads + MFMAs', 't1 epilogues', 'z epilogue', 'tail', 'loop to the next step']
print(f'{len(t)} waves; one step: {med(t[:, 6] - t[:, 0])} cycles (p10 {int(np.percentile(t[:, 6] - t[:, 0], 10))}, p90 {int(np.percentile(t[:, 6] - t[:, 0], 90))})')
for i, nm in enumerate(names):
    print(f'  {nm:40s} {med(t[:, i + 1] - t[:, i])}')
