"""How much does the second wave per SIMD add?  The streaming kernels timed at their normal residency and with LDS padding that halves it
(V2W_LDSPAD, a knob of the what-if library tools/exp/libv2w_timeline.so only).  Same grid (persistent waves walk more runs), so
time(half residency) / time(full) = 2 means the partner wave on a SIMD doubles the throughput, 1 means it adds nothing."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools", "exp"))
which = sys.argv[1]
if which == "n32s":
    import n32s_check as n
    call = n.runner(*n.make(64, 512 * 160, 7))
else:
    import n16s_check as n
    y = torch.empty((64, 1, 512 * 320), device=n.dev)
    call = n.run(*n.make(64, 512 * 320, 7), y)
best = 1e9
for rnd in range(3):
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
print(which, "LDSPAD", os.environ.get("V2W_LDSPAD", "0"), round(best, 1), "us", flush=True)
''' % (ROOT, ROOT)
env0 = dict(os.environ, V2W_LIB=os.path.join(ROOT, 'tools', 'exp', 'libv2w_timeline.so'))
for which, pads in (('n32s', (0, 30000, 110000)), ('n16s', (0, 19000, 33000, 60000))):     # n32s: 2 / 1 / 1 teams per CU; n16s: 7 / 4 / 3 (3 of the 4 SIMDs) / 2 ... waves per CU
    for pad in pads:
        env = dict(env0)
        if pad:
            env['V2W_LDSPAD'] = str(pad)
        subprocess.run([sys.executable, '-c', code, which], env=env)
