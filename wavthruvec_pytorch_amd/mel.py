"""`mel_spectrogram` of the generated audio on the MI355X.

Mirror of the reference's `dataset.mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False)`
(vec2wav/dataset.py:53-77; call sites train.py:172-174, 266-269, 282-284): same name, argument order and result
`(B, num_mels, frames)` = log(clamp(mel_basis @ |STFT|, 1e-5)), differentiable with respect to y (the training loss
back-propagates through it, train.py:204).

The STFT runs as ONE fused Conv1d on the f32 MFMA tile kernel: the reflect-padded signal is de-interleaved by hop phase
(`v2w_mel_phases`), the windowed DFT rows are the conv weights (hop input channels x n_fft/hop taps), and `v2w_mel_finish`
does magnitude -> filterbank -> log.  Constants (DFT rows, hann window, Slaney mel filterbank = librosa.filters.mel of the
reference's librosa, restated because librosa is not a dependency here) are built once per configuration with numpy in fp64.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from . import _hip, hipops

_CONSTS: Dict[Tuple, dict] = {}


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    logstep = np.log(6.4) / 27.0
    return np.where(f >= 1000.0, 1000.0 / f_sp + np.log(np.maximum(f, 1e-30) / 1000.0) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 1000.0 / f_sp, 1000.0 * np.exp(logstep * (m - 1000.0 / f_sp)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """(n_mels, n_fft//2 + 1) float32: triangular filters on the Slaney mel scale with area normalisation, the published
    algorithm of librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney') (dataset.py:9,64)."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper)) * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def _constants(n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, device):
    key = (n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, str(device))
    c = _CONSTS.get(key)
    if c is None:
        if n_fft % hop_size or hop_size % 32 or win_size != n_fft:
            raise NotImplementedError('mel_spectrogram (HIP): needs win_size == n_fft, n_fft % hop_size == 0 and hop_size % 32 == 0 '
                                      '(the reference configuration: 1024 / 256 / 1024)')
        nb, k = n_fft // 2 + 1, n_fft // hop_size
        cs = (2 * nb + 63) // 64 * 64                           # DFT rows padded to the conv tile's row block
        t = np.arange(n_fft, dtype=np.float64)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * t / win_size)    # torch.hann_window(win_size): periodic
        ang = 2.0 * np.pi * np.outer(np.arange(nb, dtype=np.float64), t) / n_fft
        wd = np.zeros((cs, n_fft), dtype=np.float64)
        wd[:nb] = np.cos(ang) * win
        wd[nb:2 * nb] = -np.sin(ang) * win
        # conv weights [tap j][channel p][row c] = Wd[c][j*hop + p]
        wf = torch.from_numpy(np.ascontiguousarray(wd.reshape(cs, k, hop_size).transpose(1, 2, 0)).astype(np.float32)).to(device)
        c = dict(nb=nb, k=k, cs=cs, wf=wf, wp=hipops.pack_mfma(wf),
                 basis=torch.from_numpy(mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)).to(device))
        _CONSTS[key] = c
    return c


def _forward(y, cfg):
    """-> (out, spec, geometry): the three launches of the forward; `spec` is what the backward needs."""
    n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax = cfg
    B, L = y.shape
    c = _constants(n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, y.device)
    pad = int((n_fft - hop_size) / 2)
    if pad >= L:
        raise RuntimeError('mel_spectrogram: reflect padding needs more than (n_fft - hop)/2 samples')
    F = (L + 2 * pad - n_fft) // hop_size + 1
    FP = (F + c['k'] - 1 + 3) // 4 * 4
    lib, st = _hip.load(), torch.cuda.current_stream(y.device).cuda_stream
    xp = torch.empty((B, hop_size, FP), device=y.device)
    _hip.check(lib.v2w_mel_phases(y.data_ptr(), xp.data_ptr(), B, L, hop_size, pad, FP, st), 'v2w_mel_phases')
    spec = torch.empty((B, c['cs'], FP), device=y.device)
    hipops.conv1d(xp, c['wf'], None, spec, k=c['k'], dil=1, slope=1.0, pad_left=0, wp=c['wp'])
    out = torch.empty((B, num_mels, F), device=y.device)
    _hip.check(lib.v2w_mel_finish(spec.data_ptr(), c['basis'].data_ptr(), out.data_ptr(), B, c['cs'], FP, F, c['nb'], num_mels, st),
               'v2w_mel_finish')
    return out, spec, (B, L, pad, F, FP)


def _backward(g, spec, geom, cfg):
    """d out (B, num_mels, F) -> d y (B, L): finish^T -> the DFT conv's input gradient (the same MFMA conv kernel with the
    tap-flipped transposed rows, pad_left = k-1) -> phase re-interleave with the reflections folded back."""
    n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax = cfg
    B, L, pad, F, FP = geom
    dev = spec.device
    c = _constants(n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, dev)
    if 'wT' not in c:
        # wT[j'][row c][phase p] = Wd[c][(k-1-j')*hop + p]
        c['wT'] = c['wf'].flip(0).transpose(1, 2).contiguous()
        c['wTp'] = hipops.pack_mfma(c['wT'])
        c['basisT'] = c['basis'].t().contiguous()
    lib, st = _hip.load(), torch.cuda.current_stream(dev).cuda_stream
    g = g.contiguous().float()
    dspec = torch.zeros_like(spec)
    _hip.check(lib.v2w_mel_finish_bwd(spec.data_ptr(), c['basis'].data_ptr(), c['basisT'].data_ptr(), g.data_ptr(), dspec.data_ptr(),
                                      B, c['cs'], FP, F, c['nb'], num_mels, st), 'v2w_mel_finish_bwd')
    dxp = torch.empty((B, hop_size, FP), device=dev)
    hipops.conv1d(dspec, c['wT'], None, dxp, k=c['k'], dil=1, slope=1.0, pad_left=c['k'] - 1, wp=c['wTp'])
    dy = torch.empty((B, L), device=dev)
    _hip.check(lib.v2w_mel_phases_bwd(dxp.data_ptr(), dy.data_ptr(), B, L, hop_size, pad, FP, st), 'v2w_mel_phases_bwd')
    return dy


class _MelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, cfg):
        out, spec, geom = _forward(y.detach().contiguous().float(), cfg)
        ctx.save_for_backward(spec)
        ctx.geom, ctx.cfg = geom, cfg
        return out

    @staticmethod
    @_hip.on_tensor_device
    def backward(ctx, g):
        (spec,) = ctx.saved_tensors
        return _backward(g, spec, ctx.geom, ctx.cfg), None


@_hip.on_tensor_device
def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """y (B, L) float32 on the GPU -> (B, num_mels, frames) float32; differentiable with respect to y (the training loss
    F.l1_loss(y_mel, mel_spectrogram(y_g_hat.squeeze(1), ...)), train.py:172-174,204)."""
    if center:
        raise NotImplementedError('mel_spectrogram (HIP): center=False only (the reference never passes True)')
    if not y.is_cuda:
        raise RuntimeError('mel_spectrogram runs on the MI355X HIP path only (no CPU fallback)')
    cfg = (n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax)
    if torch.is_grad_enabled() and y.requires_grad:
        return _MelFn.apply(y, cfg)
    with torch.no_grad():
        return _forward(y.detach().contiguous().float(), cfg)[0]
