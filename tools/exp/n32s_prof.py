"""Runs the 32-channel stage + upsampler kernel a few times at B x T (default 64 x 512) for rocprofv3 counter passes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import n32s_check as n  # noqa: E402

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 512)
call = n.runner(*n.make(B, T * 160, 7))
for _ in range(5):
    call()
torch.cuda.synchronize()
