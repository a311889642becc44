#!/usr/bin/env python3
"""Forward time of the MPD / MSD discriminators on (y, y_hat) at the generator's cfg2 output size (B=32 x 81920 samples)."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wavthruvec_pytorch_amd import synthetic  # noqa: E402
from wavthruvec_pytorch_amd.discriminators import MultiPeriodDiscriminator, MultiScaleDiscriminator  # noqa: E402


def flops(kind, B, T):
    """2 * MACs of both inputs through every conv (dense count of the reference's layers)."""
    tot = 0
    if kind == 'mpd':
        for p in synthetic.DEFAULT_PERIODS:
            H = -(-T // p)
            for ci, co, k, s, pad in synthetic.DISC_P_LAYERS + [synthetic.DISC_P_POST]:
                H = (H + 2 * pad - k) // s + 1
                tot += 2 * ci * co * k * H * p
    else:
        L0 = T
        for d in range(3):
            if d:
                L0 = L0 // 2 + 1
            L = L0
            for ci, co, k, s, g, pad in synthetic.DISC_S_LAYERS + [synthetic.DISC_S_POST]:
                L = (L + 2 * pad - k) // s + 1
                tot += 2 * (ci // g) * co * k * L
    return tot * B * 2


def stock_torch(kind, dev):
    """The same layer stack on stock torch.nn convs (MIOpen through PyTorch-ROCm, cudnn.benchmark as train.py:24 sets it):
    what the reference runs on this GPU.  Random weights; timing only."""
    import torch.nn as nn
    import torch.nn.functional as F

    class P(nn.Module):
        def __init__(self, period):
            super().__init__()
            self.period = period
            self.convs = nn.ModuleList([nn.Conv2d(ci, co, (k, 1), (s, 1), padding=(p, 0)) for ci, co, k, s, p in synthetic.DISC_P_LAYERS])
            self.post = nn.Conv2d(1024, 1, (3, 1), 1, padding=(1, 0))

        def forward(self, x):
            b, c, t = x.shape
            if t % self.period:
                x = F.pad(x, (0, self.period - t % self.period), 'reflect')
            x = x.view(b, c, -1, self.period)
            fm = []
            for l in self.convs:
                x = F.leaky_relu(l(x), 0.1); fm.append(x)
            return self.post(x), fm

    class S(nn.Module):
        def __init__(self):
            super().__init__()
            self.convs = nn.ModuleList([nn.Conv1d(ci, co, k, s, padding=p, groups=g) for ci, co, k, s, g, p in synthetic.DISC_S_LAYERS])
            self.post = nn.Conv1d(1024, 1, 3, 1, padding=1)

        def forward(self, x):
            fm = []
            for l in self.convs:
                x = F.leaky_relu(l(x), 0.1); fm.append(x)
            return self.post(x), fm

    if kind == 'mpd':
        ds = nn.ModuleList([P(p) for p in synthetic.DEFAULT_PERIODS]).to(dev)

        def run(y, y_hat):
            for d in ds:
                d(y); d(y_hat)
    else:
        ds = nn.ModuleList([S() for _ in range(3)]).to(dev)

        def run(y, y_hat):
            for i, d in enumerate(ds):
                if i:
                    y, y_hat = F.avg_pool1d(y, 4, 2, padding=2), F.avg_pool1d(y_hat, 4, 2, padding=2)
                d(y); d(y_hat)
    return run


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 81920
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device('cuda:0')
    y, y_hat = synthetic.make_audio_pair(B, T, seed=1, device=dev)
    for kind in ('mpd', 'msd'):
        if kind == 'mpd':
            m = MultiPeriodDiscriminator(SimpleNamespace(periods=synthetic.DEFAULT_PERIODS))
            m.load_state_dict(synthetic.make_disc_state_dict(synthetic.mpd_state_dict_spec(), seed=0))
        else:
            m = MultiScaleDiscriminator()
            m.load_state_dict(synthetic.make_disc_state_dict(synthetic.msd_state_dict_spec(), seed=0))
        m = m.to(dev).train()
        with torch.no_grad():
            for _ in range(2):
                m(y, y_hat)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                m(y, y_hat)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        fl = flops(kind, B, T)
        print(f'{kind} B={B} T={T}: {dt * 1e3:.2f} ms per forward(y, y_hat)  {fl / 1e9:.0f} GFLOP  {fl / dt / 1e12:.1f} TFLOP/s')
        if os.environ.get('V2W_STOCK', '0') == '1':
            torch.backends.cudnn.benchmark = True
            run = stock_torch(kind, dev)
            with torch.no_grad():
                for _ in range(3):
                    run(y, y_hat)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    run(y, y_hat)
                torch.cuda.synchronize()
            ds = (time.perf_counter() - t0) / steps
            print(f'{kind} stock torch.nn (MIOpen) same shapes: {ds * 1e3:.2f} ms  {fl / ds / 1e12:.1f} TFLOP/s   ratio {ds / dt:.2f}x')


if __name__ == '__main__':
    main()
