"""Timing of the n32s kernel from the product library and from a what-if build (V2W_LIB=tools/exp/libv2w_timeline.so built with
V2W_TL_DEFS=...: results are wrong by construction, only the time is read)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools", "exp"))
import n32s_check as n
for B, T in ((64, 512), (32, 256)):
    call = n.runner(*n.make(B, T * 160, 7))
    best = 1e9
    for rnd in range(3):
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(os.environ.get("V2W_LIB", "product"), B, T, round(best, 1), "us", flush=True)
''' % (ROOT, ROOT)
for lib in (None, os.path.join(ROOT, 'tools', 'exp', 'libv2w_timeline.so'), None):
    env = dict(os.environ)
    env.pop('V2W_LIB', None)
    if lib:
        env['V2W_LIB'] = lib
    subprocess.run([sys.executable, '-c', code], env=env)
