#!/usr/bin/env python3
"""One full vec2wav GAN training iteration as vec2wav/train.py:160-215 runs it - generator forward, mel of the generated audio,
discriminator step (MPD + MSD on y and y_g_hat.detach(), backward, AdamW), generator step (MPD + MSD again, feature / LSGAN /
L1-mel losses, backward through the discriminators, the mel and the generator, AdamW) - entirely on the HIP path.
argv: B T steps [stock] [frozen]   ('stock': the discriminators as stock torch.nn conv stacks (MIOpen + torch autograd) for comparison;
'frozen': `with discriminators.frozen(mpd, msd)` around the G step's discriminator forwards - same trajectory, no wasted D gradients)"""
import contextlib
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from wavthruvec_pytorch_amd import Generator, synthetic  # noqa: E402
from wavthruvec_pytorch_amd.mel import mel_spectrogram  # noqa: E402
from wavthruvec_pytorch_amd import discriminators as HD  # noqa: E402


class StockP(nn.Module):
    def __init__(self, period):
        super().__init__()
        self.period = period
        wn = nn.utils.weight_norm
        self.convs = nn.ModuleList([wn(nn.Conv2d(ci, co, (k, 1), (s, 1), padding=(p, 0))) for ci, co, k, s, p in synthetic.DISC_P_LAYERS])
        self.conv_post = wn(nn.Conv2d(1024, 1, (3, 1), 1, padding=(1, 0)))

    def forward(self, x):
        b, c, t = x.shape
        if t % self.period:
            x = F.pad(x, (0, self.period - t % self.period), 'reflect')
        x = x.view(b, c, -1, self.period)
        fm = []
        for l in self.convs:
            x = F.leaky_relu(l(x), 0.1); fm.append(x)
        x = self.conv_post(x); fm.append(x)
        return torch.flatten(x, 1, -1), fm


class StockS(nn.Module):
    def __init__(self, sn=False):
        super().__init__()
        nf = nn.utils.spectral_norm if sn else nn.utils.weight_norm
        self.convs = nn.ModuleList([nf(nn.Conv1d(ci, co, k, s, padding=p, groups=g)) for ci, co, k, s, g, p in synthetic.DISC_S_LAYERS])
        self.conv_post = nf(nn.Conv1d(1024, 1, 3, 1, padding=1))

    def forward(self, x):
        fm = []
        for l in self.convs:
            x = F.leaky_relu(l(x), 0.1); fm.append(x)
        x = self.conv_post(x); fm.append(x)
        return torch.flatten(x, 1, -1), fm


class StockMulti(nn.Module):
    def __init__(self, ds, pool):
        super().__init__()
        self.discriminators = nn.ModuleList(ds)
        self.pool = pool

    def forward(self, y, y_hat):
        rs, gs, fr, fg = [], [], [], []
        for i, d in enumerate(self.discriminators):
            if self.pool and i:
                y, y_hat = F.avg_pool1d(y, 4, 2, padding=2), F.avg_pool1d(y_hat, 4, 2, padding=2)
            a, b = d(y); c, e = d(y_hat)
            rs.append(a); fr.append(b); gs.append(c); fg.append(e)
        return rs, gs, fr, fg


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    stock = len(sys.argv) > 4 and sys.argv[4] == 'stock'
    freeze = 'frozen' in sys.argv[4:]      # discriminator parameters do not require grad during the G step (their grads are discarded anyway)
    if 'pairs' in sys.argv[4:]:            # (y, y_hat) of the weight-normed discriminators as one batched call at any size ('auto': small batches only)
        HD.BATCH_PAIRS = True
    if 'nopairs' in sys.argv[4:]:
        HD.BATCH_PAIRS = False
    f16x3 = 'f16x3' in sys.argv[4:]        # the dense five-tap 1024 -> 1024 discriminator convs (forward + input gradient) and the generator's
    #                                        forward / input-gradient convs as f16 hi + lo operands; every weight gradient exact
    dev = torch.device('cuda:0')
    torch.backends.cudnn.benchmark = os.environ.get("V2W_CUDNN_BENCHMARK", "1") == "1"          # train.py:24 sets True
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).train()
    if stock:
        mpd = StockMulti([StockP(p) for p in synthetic.DEFAULT_PERIODS], False).to(dev)
        msd = StockMulti([StockS(True), StockS(), StockS()], True).to(dev)
    else:
        mpd = HD.MultiPeriodDiscriminator(SimpleNamespace(periods=synthetic.DEFAULT_PERIODS))
        mpd.load_state_dict(synthetic.make_disc_state_dict(synthetic.mpd_state_dict_spec(), seed=0))
        msd = HD.MultiScaleDiscriminator()
        msd.load_state_dict(synthetic.make_disc_state_dict(synthetic.msd_state_dict_spec(), seed=0))
        mpd, msd = mpd.to(dev).train(), msd.to(dev).train()
        if f16x3:
            HD.set_precision(mpd, 'f16x3'); HD.set_precision(msd, 'f16x3')
            g.precision = 'f16x3'
    optim_g = torch.optim.AdamW(g.parameters(), 2e-4, betas=(0.8, 0.99))
    optim_d = torch.optim.AdamW(list(mpd.parameters()) + list(msd.parameters()), 2e-4, betas=(0.8, 0.99))
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    y = torch.tanh(torch.randn(B, 1, T * 320, device=dev)) * 0.5
    margs = (h.n_fft, h.num_mels, h.sampling_rate, h.hop_size, h.win_size, h.fmin, h.fmax_for_loss)
    y_mel = mel_spectrogram(y.squeeze(1), *margs)
    times = {}

    peaks, live = {}, {}

    def tick(name, t0):
        torch.cuda.synchronize()
        times[name] = times.get(name, 0.0) + time.perf_counter() - t0
        peaks[name] = max(peaks.get(name, 0), torch.cuda.max_memory_allocated())      # per phase: reset below
        live[name] = torch.cuda.memory_allocated()
        torch.cuda.reset_peak_memory_stats()
        return time.perf_counter()

    for it in range(steps + 2):          # two untimed iterations: the caching allocator reaches its working set (tens of GB)
        if it == 2:
            times.clear()
            torch.cuda.synchronize(); t_all = time.perf_counter()
            if os.environ.get('V2W_CPROFILE'):      # diagnostic: host profile of the timed iterations only
                import cProfile
                prof = cProfile.Profile(); prof.enable()
        t0 = time.perf_counter()
        y_g_hat = g(*inp)
        y_g_hat_mel = mel_spectrogram(y_g_hat.squeeze(1), *margs)
        t0 = tick('generator forward + mel', t0)
        optim_d.zero_grad()
        y_df_hat_r, y_df_hat_g, _, _ = mpd(y, y_g_hat.detach())
        loss_disc_f, _, _ = HD.discriminator_loss(y_df_hat_r, y_df_hat_g)
        y_ds_hat_r, y_ds_hat_g, _, _ = msd(y, y_g_hat.detach())
        loss_disc_s, _, _ = HD.discriminator_loss(y_ds_hat_r, y_ds_hat_g)
        t0 = tick('D step: forwards', t0)
        (loss_disc_s + loss_disc_f).backward()
        optim_d.step()
        t0 = tick('D step: backward + AdamW', t0)
        optim_g.zero_grad()
        loss_mel = F.l1_loss(y_mel, y_g_hat_mel) * 45
        with (HD.frozen(mpd, msd) if freeze else contextlib.nullcontext()):
            y_df_hat_r, y_df_hat_g, fmap_f_r, fmap_f_g = mpd(y, y_g_hat)
            y_ds_hat_r, y_ds_hat_g, fmap_s_r, fmap_s_g = msd(y, y_g_hat)
        loss_gen_all = HD.generator_loss(y_ds_hat_g)[0] + HD.generator_loss(y_df_hat_g)[0] + HD.feature_loss(fmap_s_r, fmap_s_g) \
            + HD.feature_loss(fmap_f_r, fmap_f_g) + loss_mel
        t0 = tick('G step: discriminator forwards + losses', t0)
        loss_gen_all.backward()
        optim_g.step()
        t0 = tick('G step: backward (D, mel, G) + AdamW', t0)
    torch.cuda.synchronize()
    if os.environ.get('V2W_CPROFILE'):
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats(os.environ['V2W_CPROFILE'] if os.environ['V2W_CPROFILE'] in ('tottime', 'cumtime') else 'tottime').print_stats(45)
    if os.environ.get('V2W_LIVE_TENSORS'):      # diagnostic: what is still allocated after an iteration (storages, largest first)
        import gc
        seen, rows = set(), []
        for o in gc.get_objects():
            try:
                if torch.is_tensor(o) and o.is_cuda:
                    st = o.untyped_storage()
                    if st.data_ptr() not in seen:
                        seen.add(st.data_ptr())
                        rows.append((st.nbytes(), tuple(o.shape), str(o.dtype), o.grad_fn is not None or o.requires_grad))
            except Exception:
                pass
        rows.sort(reverse=True)
        print(f'live storages: {len(rows)}, {sum(r[0] for r in rows) / 2 ** 30:.1f} GiB of {torch.cuda.memory_allocated() / 2 ** 30:.1f} GiB allocated')
        for r in rows[:40]:
            print(f'   {r[0] / 2 ** 20:9.1f} MiB  {r[1]}  {r[2]}  graph={r[3]}')
    dt = (time.perf_counter() - t_all) / steps
    print(f'{"stock torch discriminators" if stock else "HIP discriminators"}{" (frozen in the G step)" if freeze else ""}{" [f16x3 convs]" if f16x3 else ""}  B={B} T={T}: {dt * 1e3:.1f} ms per GAN iteration '
          f'({B * T * 320 / dt / 1e6:.2f} M samples/s trained), loss_gen {loss_gen_all.item():.4f}')
    for k, v in times.items():
        print(f'    {k:45s} {v / steps * 1e3:8.1f} ms   peak {peaks[k] / 2 ** 30:6.1f} GiB, live at its end {live[k] / 2 ** 30:6.1f} GiB')
    print(f'    peak device memory (torch allocator)          {max(peaks.values()) / 2 ** 30:8.1f} GiB')


if __name__ == '__main__':
    main()
