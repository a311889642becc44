#!/usr/bin/env python3
"""Forward + backward time of the generator at cfg2 (the generator half of a vec2wav/train.py step).
argv: B T steps precision loss   (loss 'sum' = a weighted sum of the waveform; 'mel' = train.py:172-174,204's
F.l1_loss(y_mel, mel_spectrogram(y_g_hat)) * 45 through the HIP mel_spectrogram and its backward)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wavthruvec_pytorch_amd import Generator, synthetic  # noqa: E402
from wavthruvec_pytorch_amd.mel import mel_spectrogram  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device('cuda:0')
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).train()
    g.precision = sys.argv[4] if len(sys.argv) > 4 else 'f32'
    opt = torch.optim.AdamW(g.parameters(), 2e-4, betas=(0.8, 0.99))
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    dy = torch.randn(B, 1, T * 320, device=dev)
    loss_kind = sys.argv[5] if len(sys.argv) > 5 else 'sum'
    margs = (h.n_fft, h.num_mels, h.sampling_rate, h.hop_size, h.win_size, h.fmin, h.fmax_for_loss)
    y_mel = mel_spectrogram(torch.tanh(torch.randn(B, T * 320, device=dev)) * 0.5, *margs)
    for it in range(steps + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        y = g(*inp)
        if loss_kind == 'mel':
            (torch.nn.functional.l1_loss(y_mel, mel_spectrogram(y.squeeze(1), *margs)) * 45).backward()
        else:
            (y * dy).sum().backward()
        opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    with torch.no_grad():
        g(*inp)          # (the first no-grad forward after training steps allocates its own workspace and plans: not part of the figure)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(steps):
            g(*inp)
        torch.cuda.synchronize()
        df = (time.perf_counter() - t1) / steps
    print(f'B={B} T={T} loss={loss_kind} {g.precision}: forward+backward+AdamW {dt * 1e3:.2f} ms/step ; inference-schedule forward {df * 1e3:.2f} ms ; '
          f'{B * T * 320 / dt / 1e6:.1f} M samples/s trained')


if __name__ == '__main__':
    main()
