"""ORACLE - test infrastructure only (see vec2wav_oracle.py for the rules).  CPU restatement of the reference's
MultiPeriodDiscriminator / MultiScaleDiscriminator forwards (vec2wav/models.py:158-275) as pure functions of a state_dict.

Pinned against fixtures captured from the reference modules themselves (tools/gen_disc_goldens.py -> tests/golden/disc_*.npz).
Weight norm is the legacy `torch.nn.utils.weight_norm` (w = g * v / |v| per output channel); spectral norm is the legacy hook
`torch.nn.utils.spectral_norm` (dim 0, one power iteration per training-mode forward, eps 1e-12, u/v buffers updated in place,
sigma = u . (W v); eval mode uses the stored u, v) - DiscriminatorS #0 only (models.py:250).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from wavthruvec_pytorch_amd.synthetic import DISC_P_LAYERS, DISC_P_POST, DISC_S_LAYERS, DISC_S_POST, DEFAULT_PERIODS

LRELU_SLOPE = 0.1          # models.py:9


def wn_weight(sd, name):
    """weight_norm: w = g * v / ||v||, the norm over every dim but 0."""
    v, g = sd[name + '.weight_v'], sd[name + '.weight_g']
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / n)


def sn_weight(sd, name, training: bool):
    """spectral_norm: W / sigma; in training mode one power iteration first, written back into sd's u / v (as the hook does)."""
    w = sd[name + '.weight_orig']
    u, v = sd[name + '.weight_u'], sd[name + '.weight_v']
    wm = w.reshape(w.shape[0], -1)
    if training:
        v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
        u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
        sd[name + '.weight_u'], sd[name + '.weight_v'] = u, v
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def disc_p(x, sd, prefix, period):
    """DiscriminatorP.forward (models.py:173-193): x (B, 1, T) -> (score (B, N), fmap list)."""
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), 'reflect')
        t = t + n_pad
    x = x.view(b, c, t // period, period)
    fmap = []
    for i, (_ci, _co, _k, s, p) in enumerate(DISC_P_LAYERS):
        x = F.conv2d(x, wn_weight(sd, f'{prefix}.convs.{i}'), sd[f'{prefix}.convs.{i}.bias'], stride=(s, 1), padding=(p, 0))
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv2d(x, wn_weight(sd, f'{prefix}.conv_post'), sd[f'{prefix}.conv_post.bias'], stride=1, padding=(DISC_P_POST[4], 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd_forward(sd, y, y_hat, periods=DEFAULT_PERIODS):
    """MultiPeriodDiscriminator.forward (models.py:203-216)."""
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
    for d, period in enumerate(periods):
        r, fr = disc_p(y, sd, f'discriminators.{d}', period)
        g, fg = disc_p(y_hat, sd, f'discriminators.{d}', period)
        y_d_rs.append(r); fmap_rs.append(fr); y_d_gs.append(g); fmap_gs.append(fg)
    return y_d_rs, y_d_gs, fmap_rs, fmap_gs


def disc_s(x, sd, prefix, spectral: bool, training: bool):
    """DiscriminatorS.forward (models.py:233-243).  The spectral-normed variant runs its power iteration per call."""
    wfn = (lambda n: sn_weight(sd, n, training)) if spectral else (lambda n: wn_weight(sd, n))
    fmap = []
    for i, (_ci, _co, _k, s, g, p) in enumerate(DISC_S_LAYERS):
        x = F.conv1d(x, wfn(f'{prefix}.convs.{i}'), sd[f'{prefix}.convs.{i}.bias'], stride=s, padding=p, groups=g)
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv1d(x, wfn(f'{prefix}.conv_post'), sd[f'{prefix}.conv_post.bias'], stride=1, padding=DISC_S_POST[5])
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def msd_forward(sd, y, y_hat, training: bool = True):
    """MultiScaleDiscriminator.forward (models.py:259-275): scales 1, 1/2, 1/4 through AvgPool1d(4, 2, padding=2);
    each discriminator sees y then y_hat (two power iterations per step on the spectral-normed one)."""
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
    for d in range(3):
        if d != 0:
            y = F.avg_pool1d(y, 4, 2, padding=2)
            y_hat = F.avg_pool1d(y_hat, 4, 2, padding=2)
        r, fr = disc_s(y, sd, f'discriminators.{d}', d == 0, training)
        g, fg = disc_s(y_hat, sd, f'discriminators.{d}', d == 0, training)
        y_d_rs.append(r); fmap_rs.append(fr); y_d_gs.append(g); fmap_gs.append(fg)
    return y_d_rs, y_d_gs, fmap_rs, fmap_gs
