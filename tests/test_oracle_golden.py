"""Pin the oracle (oracle/vec2wav_oracle.py) against every fixture captured from the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import vec2wav_oracle as O
from tests import golden_util
from tests.golden_util import golden_names, load_golden, case_setup, probe_summary, tol_for


def run_oracle(meta, h, sd, inp, inp2, dtype=torch.float32):
    mode = meta['mode']
    probes = {}
    extra = {}
    if mode == 'evalcal':
        O.calibrate_running_stats(sd, h, *inp, dtype=dtype)
        y, nb = O.generator_forward(sd, h, *inp, training=False, dtype=dtype, probes=probes)
    elif mode == 'eval':
        y, nb = O.generator_forward(sd, h, *inp, training=False, dtype=dtype, probes=probes)
    elif mode == 'train2':
        y1, nb1 = O.generator_forward(sd, h, *inp, training=True, dtype=dtype)
        extra['y_step1'] = y1
        O.apply_buffers(sd, nb1)
        y, nb = O.generator_forward(sd, h, *inp2, training=True, dtype=dtype, probes=probes)
    elif mode == 'train_rmwn':
        sd2 = O.remove_weight_norm_sd(sd)
        extra['keys_after_rmwn'] = list(sd2.keys())
        y, nb = O.generator_forward(sd2, h, *inp, training=True, dtype=dtype, probes=probes)
    else:
        y, nb = O.generator_forward(sd, h, *inp, training=True, dtype=dtype, probes=probes)
    O.apply_buffers(sd, nb)
    return y, probes, extra


@pytest.mark.parametrize('name', golden_names())
def test_oracle_matches_reference_golden(name):
    torch.set_num_threads(8)
    z, meta = load_golden(name)
    h, sd, inp, inp2 = case_setup(meta)
    y, probes, extra = run_oracle(meta, h, sd, inp, inp2)
    tol = tol_for(meta)
    # tighter than the product bar: oracle and reference are the same arithmetic up to summation order
    otol = 2e-5 if meta['mode'] != 'eval' else tol
    d = np.abs(y.numpy() - z['y']).max()
    assert y.shape == z['y'].shape
    assert d <= otol, f'{name}: max|dy|={d}'
    if 'y_step1' in extra:
        assert np.abs(extra['y_step1'].numpy() - z['y_step1']).max() <= otol
    if 'keys_after_rmwn' in extra:
        assert extra['keys_after_rmwn'] == [str(k) for k in z['keys_after_rmwn']]
    # per-layer probes
    for pname, t in probes.items():
        s = probe_summary(t)
        scale = max(1.0, float(np.abs(z[f'probe/{pname}/head']).max()))
        ptol = (1e-3 if meta['mode'] != 'eval' else 5e-2) * scale
        assert np.abs(s['head'] - z[f'probe/{pname}/head']).max() <= ptol, pname
        assert np.abs(s['tail'] - z[f'probe/{pname}/tail']).max() <= ptol, pname
        ref_abs = float(z[f'probe/{pname}/abssum'])
        assert abs(s['abssum'] - ref_abs) <= 1e-4 * ref_abs + 1e-3, pname
    # post-forward buffers (running stats, num_batches_tracked, spectral-norm u/v)
    for k in z.files:
        if k.startswith('buf/'):
            ref = z[k]
            got = sd[k[4:]].numpy()
            assert got.shape == ref.shape, k
            if ref.dtype.kind == 'i':
                assert (got == ref).all(), k
            else:
                assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k


def test_state_dict_spec_matches_reference_keys():
    from wavthruvec_pytorch_amd import synthetic
    z, meta = load_golden('rb2_train_b2_t8')
    h = synthetic.make_hparams(**meta['hp'])
    spec = synthetic.state_dict_spec(h)
    assert [k for k, _, _ in spec] == [str(k) for k in z['meta_keys']]
    assert [','.join(map(str, s)) for _, s, _ in spec] == [str(s) for s in z['meta_shapes']]
    z1, meta1 = load_golden('rb1_train_b2_t8')
    h1 = synthetic.make_hparams(**meta1['hp'])
    assert [k for k, _, _ in synthetic.state_dict_spec(h1)] == [str(k) for k in z1['meta_keys']]


def test_fp64_oracle_noise_floor():
    """fp32 oracle vs fp64 oracle in train mode: the path's own noise floor is ~1e-6."""
    z, meta = load_golden('rb2_train_b2_t8')
    h, sd, inp, _ = case_setup(meta)
    y32, _ = O.generator_forward(sd, h, *inp, training=True, dtype=torch.float32)
    y64, _ = O.generator_forward(sd, h, *inp, training=True, dtype=torch.float64)
    assert (y32.double() - y64).abs().max().item() < 5e-6
    assert np.abs(y64.numpy() - z['y']).max() < 5e-6


def test_mel_oracle_stft_half_and_filterbank_properties():
    """The mel oracle's STFT half is torch.stft as in the reference (dataset.py:69-72); its frame/DFT form (what the HIP path
    computes: windowed DFT rows applied to hop-strided frames of the reflect-padded signal) must agree with it.  The Slaney
    filterbank is restated (librosa is absent: unpinned against the reference) - its published invariants are checked."""
    import numpy as np
    import torch
    from oracle import mel_oracle as M
    y = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (2, 4096)))
    n_fft, hop = 1024, 256
    pad = (n_fft - hop) // 2
    yp = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode='reflect').squeeze(1)
    ref = torch.stft(yp, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft, dtype=torch.float64), center=False,
                     onesided=True, return_complex=True)
    frames = yp.unfold(1, n_fft, hop)                                        # (B, F, n_fft)
    t = torch.arange(n_fft, dtype=torch.float64)
    win = 0.5 - 0.5 * torch.cos(2 * np.pi * t / n_fft)
    ang = 2 * np.pi * torch.outer(torch.arange(n_fft // 2 + 1, dtype=torch.float64), t) / n_fft
    re = torch.einsum('bft,ct->bcf', frames * win, torch.cos(ang))
    im = torch.einsum('bft,ct->bcf', frames * win, -torch.sin(ang))
    assert (re - ref.real).abs().max() < 1e-9 and (im - ref.imag).abs().max() < 1e-9
    fb = M.mel_filterbank(16000, 1024, 80, 0, 8000)
    assert fb.shape == (80, 513) and (fb >= 0).all()
    peaks = fb.argmax(1)
    assert (np.diff(peaks) > 0).all()                                        # centre frequencies increase
    # Slaney normalisation: every triangle has (nearly) the same area in Hz; below 1 kHz the scale is linear (equal spacing)
    area = fb.sum(1) * (8000 / 512)
    assert abs(area[10:].mean() - 1.0) < 0.02 and area[10:].std() < 0.02
    lin = peaks[: int(np.searchsorted(peaks * (8000 / 512), 1000))]
    assert np.ptp(np.diff(lin)) <= 1


@pytest.mark.parametrize('name', golden_util.disc_golden_names())
def test_disc_oracle_matches_reference_goldens(name):
    """oracle/disc_oracle.py (MPD / MSD restated as functions of a state_dict) against the fixtures captured from the reference
    modules (tools/gen_disc_goldens.py): every score, every feature map's probes, the spectral-norm buffers after the forward."""
    from oracle import disc_oracle as D
    z, meta = golden_util.load_golden(name)
    sd, y, y_hat = golden_util.disc_case_setup(meta)
    with torch.no_grad():
        if meta['kind'] == 'mpd':
            outs = D.mpd_forward(sd, y, y_hat)
        else:
            if meta['mode'] == 'traineval':
                D.msd_forward(sd, y_hat, y, training=True)
            outs = D.msd_forward(sd, y, y_hat, training=meta['mode'] == 'train')
    golden_util.check_disc_outputs(z, outs, 2e-5)
    for k in z.files:
        if k.startswith('buf_'):
            assert np.abs(sd[k[4:]].numpy() - z[k]).max() <= 1e-6, k


@pytest.mark.parametrize('name', golden_util.disc_grad_golden_names())
def test_disc_oracle_autograd_matches_reference_gradients(name):
    """torch autograd through oracle/disc_oracle.py against the gradients captured from the reference modules' own backward
    (feature + generator + discriminator losses; parameters incl. the spectral-normed weight_orig, and dL/dy_hat)."""
    from oracle import disc_oracle as D
    z, meta = golden_util.load_golden(name)
    sd, y, y_hat = golden_util.disc_case_setup(meta)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not (k.endswith('weight_u') or (k.endswith('weight_v') and k.replace('weight_v', 'weight_orig') in sd))}
    sdo = dict(sd); sdo.update(leaves)
    y_hat = y_hat.clone().requires_grad_(True)
    outs = D.mpd_forward(sdo, y, y_hat) if meta['kind'] == 'mpd' else D.msd_forward(sdo, y, y_hat, training=True)
    D.mixed_loss(outs).backward()
    golden_util.check_disc_grads(z, [(k, v.grad) for k, v in leaves.items()], y_hat.grad, rtol=1e-3)   # sign() of the L1 feature loss: a few entries flip


def test_mel_filterbank_pinned_to_independent_implementation_and_librosa_docs(golden_dir):
    """`librosa.filters.mel` (dataset.py:9,64) is restated twice here - oracle/mel_oracle.py (checker) and
    wavthruvec_pytorch_amd/mel.py (product) - because librosa is not installed.  Both are held to tests/golden/mel_filterbank.npz
    (tools/gen_mel_goldens.py): the matrices of an independent third-party implementation (transformers.audio_utils.mel_filter_bank,
    Slaney scale + Slaney norm) for the reference's configuration and for librosa's documentation example, and the numeric rows
    printed in librosa's documentation.  A wrong Slaney constant (200/3 Hz per mel, 1 kHz knee, log(6.4)/27 step, 2/bandwidth norm)
    fails here."""
    import numpy as np
    from oracle import mel_oracle as M
    from wavthruvec_pytorch_amd import mel as P
    z = np.load(os.path.join(golden_dir, 'mel_filterbank.npz'))
    names = [k[len('basis.'):] for k in z.files if k.startswith('basis.')]
    assert 'ref_16k_1024_80_0_8000' in names and len(names) >= 2
    for n in names:
        sr, n_fft, n_mels, fmin, fmax = z['cfg.' + n]
        want = z['basis.' + n]
        for impl in (M.mel_filterbank, P.mel_filterbank):
            got = impl(int(sr), int(n_fft), int(n_mels), float(fmin), float(fmax))
            assert got.shape == want.shape and got.dtype == np.float32
            assert np.abs(got - want).max() <= 1e-8, (n, impl.__module__)          # entries are <= 0.04: ~1 ulp of float32
    for mod in (M, P):
        for hz, mel in z['doc.hz_to_mel']:
            assert abs(float(mod._hz_to_mel(hz)) - mel) < 5e-3
        for mel, hz in z['doc.mel_to_hz']:
            assert abs(float(mod._mel_to_hz(mel)) - hz) < 5e-4
        f40 = mod._mel_to_hz(np.linspace(mod._hz_to_mel(0.0), mod._hz_to_mel(11025.0), 40))
        assert np.abs(f40 - z['doc.mel_frequencies_40']).max() < 5e-4              # printed to 3 decimals
        row0 = mod.mel_filterbank(22050, 2048, 128)[0, :2]
        assert np.abs(np.round(row0, 3) - z['doc.filters_mel_22050_2048_row0']).max() < 1e-9
