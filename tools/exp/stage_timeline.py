import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wavthruvec_pytorch_amd import _hip
os.environ.setdefault('X', '1')
import runpy
sys.argv = ['x']
runpy.run_path(os.path.join(os.path.dirname(__file__), '..', 'stage_split_microbench.py'))
f = ctypes.CDLL(_hip.lib_path()).v2w_stage_debug_read; f.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(256, dtype=np.uint64); f(buf.ctypes.data, buf.nbytes)
t = buf[:16].astype(np.int64)
names = ['prologue (X staging)'] + sum([[f'b{j} conv1', f'b{j} epi1+barrier', f'b{j} conv2'] for j in range(3)], [])
print('last launch (bf16, C=16) of workgroup 700, wave 0, cycles:')
print('  prologue', t[1] - t[0])
for j in range(3):
    print(f'  branch {j}: gap {t[2 + 4 * j] - (t[1] if j == 0 else t[5 + 4 * (j - 1)])}  conv1 {t[3 + 4 * j] - t[2 + 4 * j]}  epi1 {t[4 + 4 * j] - t[3 + 4 * j]}  conv2 {t[5 + 4 * j] - t[4 + 4 * j]}')
print('  epi2 of last branch', t[14] - t[13], ' store', t[15] - t[14], ' total', t[15] - t[0])
