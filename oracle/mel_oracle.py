"""ORACLE - test infrastructure only (see vec2wav_oracle.py for the rules).  CPU restatement of the reference's
`mel_spectrogram` (vec2wav/dataset.py:53-77): reflect pad (n_fft - hop)/2 -> STFT (hann, center=False, onesided) ->
sqrt(re^2 + im^2 + 1e-9) -> mel filterbank -> log(clamp(., 1e-5)) (dataset.py:31-41).

Pinning.  The STFT half is stock `torch.stft` here exactly as in the reference (tests compare the frame/DFT restatement used by
the HIP path against it).  The mel filterbank is `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` in the reference
(dataset.py:9,64; librosa is NOT installed in the build container and the reference module cannot be imported without it), so
`mel_filterbank` restates librosa's published algorithm (Slaney scale, htk=False, norm='slaney'; librosa 0.8/0.9
`filters.mel` / `mel_frequencies` / `hz_to_mel`): PARITY OF THE FILTERBANK IS UNPINNED against the reference.
"""
from __future__ import annotations

import numpy as np
import torch


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """(n_mels, 1 + n_fft//2) float32 - librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney')."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def mel_spectrogram(y: torch.Tensor, n_fft=1024, num_mels=80, sampling_rate=16000, hop_size=256, win_size=1024, fmin=0, fmax=None,
                    center=False, dtype=torch.float32) -> torch.Tensor:
    """y (B, L) in [-1, 1] -> (B, num_mels, frames); line by line dataset.py:53-77."""
    y = y.to(dtype)
    mel = torch.from_numpy(mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)).to(dtype)
    window = torch.hann_window(win_size, dtype=dtype)
    pad = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode='reflect').squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=window, center=center, pad_mode='reflect',
                      normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)
    spec = torch.matmul(mel, spec)
    return torch.log(torch.clamp(spec, min=1e-5))
