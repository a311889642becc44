"""Runs the 16-channel stage + tail kernel a few times at B x T (default 64 x 512) for rocprofv3 counter passes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import n16s_check as n  # noqa: E402

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 512)
L = T * 320
args = n.make(B, L, 7)
y = torch.empty((B, 1, L), device=n.dev)
call = n.run(*args, y)
for _ in range(5):
    call()
torch.cuda.synchronize()
