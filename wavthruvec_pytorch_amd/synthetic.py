"""Portable synthetic weights and inputs for the Vec2Wav generator path.

There is no network on the build or GPU boxes, so neither trained checkpoints nor
wav2vec-2.0 latents exist.  Everything that needs "a generator with weights" (the
golden fixtures, the parity tests, ``bench.py``, ``__graft_entry__.smoke``) draws
them from here: ``numpy.random.default_rng(seed)`` (PCG64, bit-stable across
platforms and numpy versions) filling every ``state_dict`` key of the reference
``Generator`` in the reference's own key order
(/root/reference/vec2wav/models.py:78-114 defines the modules;
SURVEY.md section 8(b) lists the keys), so 34 MB of weights never have to be committed.

The scales are chosen so that activations stay O(1) through all five stages
(a fresh ``torch`` init is ill-conditioned in eval mode):

  conv ``weight_v``           U(-a, a), a = 1/sqrt(fan_in)
  conv ``weight_g``           ||v|| * (1 + 0.1 N(0,1))   (norm over the weight-norm group)
  biases                      0.05 N(0,1)
  ``cbns.i.layer.weight_orig``  N(1, 0.02)              (modules.py:17)
  ``cbns.i.layer.weight_u/_v``  unit-norm gaussian vectors
  ``running_mean``            0.1 N(0,1)
  ``running_var``             U(0.5, 1.5)
  ``fcs.i.weight``            U(-a, a), a = 1/sqrt(384)
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Tuple

import numpy as np

# Generator-side defaults of /root/reference/vec2wav/hparams.py:25-27,30,40-44,51.
DEFAULT_HPARAMS = dict(
    num_wv_feat=1024,
    spk_dim=192,
    noise_dim=192,
    resblock=1,  # int 1 != '1'  ->  ResBlock2
    upsample_rates=[5, 4, 4, 2, 2],
    upsample_kernel_sizes=[11, 8, 8, 4, 4],
    upsample_initial_channel=512,
    resblock_kernel_sizes=[3, 7, 11],
    resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    # audio front end read by mel_spectrogram's call sites (hparams.py:50-61)
    num_mels=80, n_fft=1024, hop_size=256, win_size=1024, sampling_rate=16000, fmin=0, fmax=8000, fmax_for_loss=None,
)


def make_hparams(**overrides) -> SimpleNamespace:
    """Attribute bag with the nine attributes ``Generator(h)`` reads (models.py:81-84,89-97,110)."""
    d = dict(DEFAULT_HPARAMS)
    d.update(overrides)
    return SimpleNamespace(**d)


def uses_resblock1(h) -> bool:
    # models.py:84 compares against the *string* '1'.
    return h.resblock == '1'


def state_dict_spec(h, weight_norm: bool = True) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (key, shape, kind) list of the reference Generator's ``state_dict``.

    kind in {conv_bias, conv_g, conv_v, convt_g, convt_v, conv_w, convt_w, rmean, rvar, nbt,
    sn_bias, sn_w, sn_u, sn_v, fc_w, fc_b}.  With ``weight_norm=False`` the keys are those
    after ``Generator.remove_weight_norm()`` (models.py:149-156).
    """
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    c0 = h.upsample_initial_channel

    def conv(prefix, cout, cin, k, transposed=False):
        spec.append((prefix + '.bias', (cout,), 'conv_bias'))
        wshape = (cin, cout, k) if transposed else (cout, cin, k)
        if weight_norm:
            spec.append((prefix + '.weight_g', (wshape[0], 1, 1), 'convt_g' if transposed else 'conv_g'))
            spec.append((prefix + '.weight_v', wshape, 'convt_v' if transposed else 'conv_v'))
        else:
            spec.append((prefix + '.weight', wshape, 'convt_w' if transposed else 'conv_w'))

    conv('conv_pre', c0, h.num_wv_feat, 7)
    for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
        conv(f'ups.{i}', c0 // 2 ** (i + 1), c0 // 2 ** i, k, transposed=True)
    nk = len(h.resblock_kernel_sizes)
    ch = c0
    for i in range(len(h.upsample_rates)):
        ch = c0 // 2 ** (i + 1)
        for j, (k, d) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
            p = f'resblocks.{i * nk + j}'
            if uses_resblock1(h):
                for n in range(3):
                    conv(f'{p}.convs1.{n}', ch, ch, k)
                for n in range(3):
                    conv(f'{p}.convs2.{n}', ch, ch, k)
            else:
                for n in range(2):
                    conv(f'{p}.convs.{n}', ch, ch, k)
    conv('conv_post', 1, ch, 7)
    for i in range(len(h.upsample_rates)):
        c = 256 // 2 ** i  # hard-coded in models.py:113
        p = f'cbns.{i}'
        spec.append((p + '.batch_nrom.running_mean', (c,), 'rmean'))
        spec.append((p + '.batch_nrom.running_var', (c,), 'rvar'))
        spec.append((p + '.batch_nrom.num_batches_tracked', (), 'nbt'))
        spec.append((p + '.layer.bias', (2 * c,), 'sn_bias'))
        spec.append((p + '.layer.weight_orig', (2 * c, 128), 'sn_w'))
        spec.append((p + '.layer.weight_u', (2 * c,), 'sn_u'))
        spec.append((p + '.layer.weight_v', (128,), 'sn_v'))
    for i in range(len(h.upsample_rates)):
        spec.append((f'fcs.{i}.weight', (128, h.spk_dim + h.noise_dim), 'fc_w'))
        spec.append((f'fcs.{i}.bias', (128,), 'fc_b'))
    return spec


def make_state_dict_numpy(h, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic weights for every key of ``state_dict_spec(h)`` (weight-normed form)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    pending_g = None  # (key, shape) of a weight_g waiting for its weight_v
    for key, shape, kind in state_dict_spec(h, weight_norm=True):
        if kind in ('conv_g', 'convt_g'):
            pending_g = (key, shape)
            sd[key] = None  # keep key order; filled when v is known
            continue
        if kind in ('conv_v', 'convt_v'):
            # fan_in of the op: Conv1d (cout,cin,k) -> cin*k ; ConvTranspose1d (cin,cout,k): each output
            # sample sums cin*k/stride products, use cin*k/2 as a middle scale so activations stay O(1).
            if kind == 'conv_v':
                fan_in = shape[1] * shape[2]
            else:
                fan_in = shape[0] * shape[2] / 2.0
            a = 1.0 / math.sqrt(fan_in)
            v = rng.uniform(-a, a, size=shape).astype(np.float32)
            norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True))
            gkey, gshape = pending_g
            g = norm * (1.0 + 0.1 * rng.standard_normal(gshape))
            sd[gkey] = g.astype(np.float32)
            sd[key] = v
            pending_g = None
        elif kind in ('conv_bias', 'sn_bias', 'fc_b'):
            sd[key] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif kind == 'rmean':
            sd[key] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif kind == 'rvar':
            sd[key] = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif kind == 'nbt':
            sd[key] = np.array(0, dtype=np.int64)
        elif kind == 'sn_w':
            sd[key] = (1.0 + 0.02 * rng.standard_normal(shape)).astype(np.float32)
        elif kind in ('sn_u', 'sn_v'):
            x = rng.standard_normal(shape)
            sd[key] = (x / np.linalg.norm(x)).astype(np.float32)
        elif kind == 'fc_w':
            a = 1.0 / math.sqrt(shape[1])
            sd[key] = rng.uniform(-a, a, size=shape).astype(np.float32)
        else:  # pragma: no cover
            raise AssertionError(kind)
    return sd


def make_state_dict(h, seed: int = 0, device='cpu'):
    """``make_state_dict_numpy`` as an ordered dict of torch tensors."""
    import torch
    from collections import OrderedDict
    out = OrderedDict()
    for k, v in make_state_dict_numpy(h, seed).items():
        v = np.asarray(v)
        t = torch.from_numpy(np.ascontiguousarray(v)).reshape(v.shape)  # ascontiguousarray promotes 0-d to 1-d
        out[k] = t.to(device)
    return out


def make_inputs_numpy(h, batch: int, n_frame: int, seed: int = 1234):
    """``x (B, num_wv_feat, T)`` channels-first (dataset.py:212-213), ``spk_emb (B,192)``, ``noise (B,192)``; all N(0,1) fp32."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, h.num_wv_feat, n_frame), dtype=np.float32)
    spk = rng.standard_normal((batch, h.spk_dim), dtype=np.float32)
    noise = rng.standard_normal((batch, h.noise_dim), dtype=np.float32)
    return x, spk, noise


def make_inputs(h, batch: int, n_frame: int, seed: int = 1234, device='cpu'):
    import torch
    return tuple(torch.from_numpy(a).to(device) for a in make_inputs_numpy(h, batch, n_frame, seed))


def total_upsample(h) -> int:
    r = 1
    for u in h.upsample_rates:
        r *= int(u)
    return r


# ---------------------------------------------------------------------------------------------------------------------
# MPD / MSD discriminators (models.py:158-275): state_dict surface and deterministic weights

# DiscriminatorP conv stack (models.py:164-170): (C_in, C_out, k, stride, pad) along the folded time axis
DISC_P_LAYERS = [(1, 32, 5, 3, 2), (32, 128, 5, 3, 2), (128, 512, 5, 3, 2), (512, 1024, 5, 3, 2), (1024, 1024, 5, 1, 2)]
DISC_P_POST = (1024, 1, 3, 1, 1)
# DiscriminatorS conv stack (models.py:220-229): (C_in, C_out, k, stride, groups, pad)
DISC_S_LAYERS = [(1, 128, 15, 1, 1, 7), (128, 128, 41, 2, 4, 20), (128, 256, 41, 2, 16, 20), (256, 512, 41, 4, 16, 20),
                 (512, 1024, 41, 4, 16, 20), (1024, 1024, 41, 1, 16, 20), (1024, 1024, 5, 1, 1, 2)]
DISC_S_POST = (1024, 1, 3, 1, 1, 1)
DEFAULT_PERIODS = [13, 17, 19]     # hparams.py:46


def mpd_state_dict_spec(periods=DEFAULT_PERIODS) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (key, shape, kind) of MultiPeriodDiscriminator.state_dict(): weight-normed Conv2d with (k, 1) kernels."""
    spec = []
    for d in range(len(periods)):
        layers = [(f'discriminators.{d}.convs.{i}', L) for i, L in enumerate(DISC_P_LAYERS)] + \
                 [(f'discriminators.{d}.conv_post', DISC_P_POST)]
        for name, (ci, co, k, _s, _p) in layers:
            spec.append((name + '.bias', (co,), 'conv_bias'))
            spec.append((name + '.weight_g', (co, 1, 1, 1), 'conv_g'))
            spec.append((name + '.weight_v', (co, ci, k, 1), 'conv_v'))
    return spec


def msd_state_dict_spec() -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (key, shape, kind) of MultiScaleDiscriminator.state_dict(): discriminator 0 is spectral-normed
    (bias, weight_orig, weight_u, weight_v), 1 and 2 weight-normed (models.py:249-253)."""
    spec = []
    for d in range(3):
        layers = [(f'discriminators.{d}.convs.{i}', L) for i, L in enumerate(DISC_S_LAYERS)] + \
                 [(f'discriminators.{d}.conv_post', DISC_S_POST)]
        for name, (ci, co, k, _s, g, _p) in layers:
            spec.append((name + '.bias', (co,), 'conv_bias'))
            if d == 0:
                spec.append((name + '.weight_orig', (co, ci // g, k), 'conv_w'))
                spec.append((name + '.weight_u', (co,), 'sn_u'))
                spec.append((name + '.weight_v', (ci // g * k,), 'sn_v'))
            else:
                spec.append((name + '.weight_g', (co, 1, 1), 'conv_g'))
                spec.append((name + '.weight_v', (co, ci // g, k), 'conv_v'))
    return spec


def make_disc_state_dict(spec, seed: int = 0, device='cpu'):
    """Deterministic weights for a discriminator spec: v ~ U(-a, a) with a = sqrt(3 / fan_in) * 1.4 (unit-variance-preserving
    through leaky_relu(0.1) stacks, so all fmaps stay O(1)), g = |v| * (1 + 0.1 N), small biases, unit u / v."""
    import torch
    from collections import OrderedDict
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = OrderedDict()
    pending_g = None
    for key, shape, kind in spec:
        if kind == 'conv_g':
            pending_g = (key, shape); sd[key] = None
        elif kind in ('conv_v', 'conv_w'):
            fan_in = int(np.prod(shape[1:]))
            a = 1.4 * math.sqrt(3.0 / fan_in)
            v = rng.uniform(-a, a, size=shape).astype(np.float32)
            sd[key] = v
            if kind == 'conv_v':
                gkey, gshape = pending_g
                norm = np.sqrt((v.astype(np.float64) ** 2).reshape(shape[0], -1).sum(axis=1)).reshape(gshape)
                sd[gkey] = (norm * (1.0 + 0.1 * rng.standard_normal(gshape))).astype(np.float32)
                pending_g = None
        elif kind == 'conv_bias':
            sd[key] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif kind in ('sn_u', 'sn_v'):
            x = rng.standard_normal(shape)
            sd[key] = (x / np.linalg.norm(x)).astype(np.float32)
        else:  # pragma: no cover
            raise AssertionError(kind)
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)).to(device)) for k, v in sd.items())


def make_audio_pair(batch: int, n_samples: int, seed: int = 77, device='cpu'):
    """(y, y_hat), each (B, 1, n_samples) in (-1, 1): the real / generated waveforms the discriminators compare."""
    import torch
    rng = np.random.default_rng(seed)
    y = np.tanh(rng.standard_normal((batch, 1, n_samples))).astype(np.float32) * 0.9
    y_hat = np.tanh(y + 0.3 * rng.standard_normal((batch, 1, n_samples))).astype(np.float32)
    return torch.from_numpy(y).to(device), torch.from_numpy(y_hat).to(device)
