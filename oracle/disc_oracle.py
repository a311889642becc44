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


class _LreluMasked(torch.autograd.Function):
    """leaky_relu whose DERIVATIVE mask comes from a reference tensor's sign instead of x's own: lets a test compare two fp32
    implementations' gradients exactly (an element whose pre-activation is within rounding of 0 otherwise flips the mask, and
    at small B*L one flip moves every upstream gradient by percents)."""

    @staticmethod
    def forward(ctx, x, ref):
        ctx.save_for_backward(ref > 0)
        return F.leaky_relu(x, LRELU_SLOPE)

    @staticmethod
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        return torch.where(pos, g, g * LRELU_SLOPE), None


def _act(x, masks, i):
    return F.leaky_relu(x, LRELU_SLOPE) if masks is None else _LreluMasked.apply(x, masks[i].reshape(x.shape).to(x.dtype))


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
        with torch.no_grad():        # the hook iterates under no_grad and differentiates sigma = u.(W v) with u, v constants
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
        sd[name + '.weight_u'], sd[name + '.weight_v'] = u, v
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def disc_p(x, sd, prefix, period, masks=None):
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
        x = _act(x, masks, i)
        fmap.append(x)
    x = F.conv2d(x, wn_weight(sd, f'{prefix}.conv_post'), sd[f'{prefix}.conv_post.bias'], stride=1, padding=(DISC_P_POST[4], 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd_forward(sd, y, y_hat, periods=DEFAULT_PERIODS, masks=None):
    """MultiPeriodDiscriminator.forward (models.py:203-216)."""
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
    for d, period in enumerate(periods):
        r, fr = disc_p(y, sd, f'discriminators.{d}', period, masks and masks['r'][d])
        g, fg = disc_p(y_hat, sd, f'discriminators.{d}', period, masks and masks['g'][d])
        y_d_rs.append(r); fmap_rs.append(fr); y_d_gs.append(g); fmap_gs.append(fg)
    return y_d_rs, y_d_gs, fmap_rs, fmap_gs


def disc_s(x, sd, prefix, spectral: bool, training: bool, masks=None):
    """DiscriminatorS.forward (models.py:233-243).  The spectral-normed variant runs its power iteration per call."""
    wfn = (lambda n: sn_weight(sd, n, training)) if spectral else (lambda n: wn_weight(sd, n))
    fmap = []
    for i, (_ci, _co, _k, s, g, p) in enumerate(DISC_S_LAYERS):
        x = F.conv1d(x, wfn(f'{prefix}.convs.{i}'), sd[f'{prefix}.convs.{i}.bias'], stride=s, padding=p, groups=g)
        x = _act(x, masks, i)
        fmap.append(x)
    x = F.conv1d(x, wfn(f'{prefix}.conv_post'), sd[f'{prefix}.conv_post.bias'], stride=1, padding=DISC_S_POST[5])
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def msd_forward(sd, y, y_hat, training: bool = True, masks=None):
    """MultiScaleDiscriminator.forward (models.py:259-275): scales 1, 1/2, 1/4 through AvgPool1d(4, 2, padding=2);
    each discriminator sees y then y_hat (two power iterations per step on the spectral-normed one)."""
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
    for d in range(3):
        if d != 0:
            y = F.avg_pool1d(y, 4, 2, padding=2)
            y_hat = F.avg_pool1d(y_hat, 4, 2, padding=2)
        r, fr = disc_s(y, sd, f'discriminators.{d}', d == 0, training, masks and masks['r'][d])
        g, fg = disc_s(y_hat, sd, f'discriminators.{d}', d == 0, training, masks and masks['g'][d])
        y_d_rs.append(r); fmap_rs.append(fr); y_d_gs.append(g); fmap_gs.append(fg)
    return y_d_rs, y_d_gs, fmap_rs, fmap_gs


def feature_loss(fmap_r, fmap_g):
    """models.py:278-284."""
    loss = 0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(rl - gl))
    return loss * 2


def mixed_loss(outs):
    """feature_loss + generator_loss + discriminator_loss (models.py:278-310) in one scalar: every output of a
    discriminator forward gets a gradient (the backward fixtures and tests use it)."""
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = outs
    loss = feature_loss(fmap_rs, fmap_gs)
    for dr, dg in zip(y_d_rs, y_d_gs):
        loss = loss + torch.mean((1 - dg) ** 2) + torch.mean((1 - dr) ** 2) + torch.mean(dg ** 2)
    return loss


def smooth_loss(outs):
    """A differentiable-everywhere stand-in for `mixed_loss` (squared feature differences instead of |.|): exact gradient tests."""
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = outs
    loss = 0
    for dr, dg in zip(fmap_rs, fmap_gs):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean((rl - gl) ** 2) * 20
    for dr, dg in zip(y_d_rs, y_d_gs):
        loss = loss + torch.mean((1 - dg) ** 2) + torch.mean((1 - dr) ** 2) + torch.mean(dg ** 2)
    return loss
