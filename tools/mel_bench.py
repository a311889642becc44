#!/usr/bin/env python3
"""mel_spectrogram (dataset.py:53-77) of the cfg2 generator output (B=32, 81 920 samples each) on the HIP path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavthruvec_pytorch_amd.mel import mel_spectrogram
dev = torch.device('cuda:0')
y = torch.tanh(torch.randn(32, 81920, device=dev))
for _ in range(3): m = mel_spectrogram(y, 1024, 80, 16000, 256, 1024, 0, None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m = mel_spectrogram(y, 1024, 80, 16000, 256, 1024, 0, None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 2.0 * 32 * 320 * 1024 * 1026
print(f'mel_spectrogram B=32 L=81920 -> {tuple(m.shape)}: {ms:.3f} ms ({fl / ms / 1e9:.1f} TFLOP/s on the DFT, {32 * 81920 / ms / 1e3:.0f} M samples/s)')
