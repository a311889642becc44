#!/usr/bin/env python3
"""Fixtures that pin the Slaney mel filterbank of `mel_spectrogram` (vec2wav/dataset.py:9,64: `librosa.filters.mel(sampling_rate,
n_fft, num_mels, fmin, fmax)`).

librosa is not installed in the build container (no network), so the reference's own `dataset.py` cannot be imported and the
reference cannot generate this fixture itself.  Two independent sources stand in for it:

 1. `transformers.audio_utils.mel_filter_bank(..., norm='slaney', mel_scale='slaney')` - a separate third-party implementation
    that IS installed here (transformers 5.x) and that the Hugging Face test-suite holds to `librosa.filters.mel`; its matrices for
    the reference's configuration (hparams.py:50-60: 16 kHz, n_fft 1024, 80 mels, 0-8000 Hz) and for librosa's documentation
    example (22 050 Hz, n_fft 2048, 128 mels) are stored in full;
 2. the numeric rows printed in librosa's published documentation (`hz_to_mel`, `mel_to_hz`, `mel_frequencies(n_mels=40)`,
    `filters.mel(sr=22050, n_fft=2048)`), typed in below to the printed precision.

    python tools/gen_mel_goldens.py        # writes tests/golden/mel_filterbank.npz

tests/test_oracle_golden.py holds BOTH restatements (oracle/mel_oracle.py and wavthruvec_pytorch_amd/mel.py) to these.
The fixture is data (matrices and printed constants); no reference or third-party source text is stored.
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# librosa documentation, printed values (3 decimals)
DOC_HZ_TO_MEL = {60.0: 0.9, 110.0: 1.65, 220.0: 3.3, 440.0: 6.6}
DOC_MEL_TO_HZ = {1.0: 66.667, 2.0: 133.333, 3.0: 200.0, 4.0: 266.667, 5.0: 333.333}
DOC_MEL_FREQUENCIES_40 = [
    0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856, 1119.114,
    1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799, 3216.731,
    3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107, 8467.272, 9246.028,
    10096.408, 11025.]
DOC_FILTERS_MEL_22050_2048_ROW0 = [0., 0.016]      # melfb[0, :2] of librosa.filters.mel(sr=22050, n_fft=2048)

CONFIGS = {                                         # name: (sr, n_fft, n_mels, fmin, fmax)
    'ref_16k_1024_80_0_8000': (16000, 1024, 80, 0.0, 8000.0),       # vec2wav/hparams.py:50-60
    'doc_22050_2048_128': (22050, 2048, 128, 0.0, 11025.0),         # librosa's documentation example
}


def main():
    from transformers.audio_utils import mel_filter_bank
    import transformers
    out = {'transformers_version': np.array(transformers.__version__)}
    for name, (sr, n_fft, n_mels, fmin, fmax) in CONFIGS.items():
        fb = mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=fmin, max_frequency=fmax,
                             sampling_rate=sr, norm='slaney', mel_scale='slaney')
        out['basis.' + name] = np.ascontiguousarray(fb.T).astype(np.float32)          # (n_mels, 1 + n_fft/2) as librosa returns it
        out['cfg.' + name] = np.array([sr, n_fft, n_mels, fmin, fmax], dtype=np.float64)
    out['doc.hz_to_mel'] = np.array(sorted(DOC_HZ_TO_MEL.items()), dtype=np.float64)
    out['doc.mel_to_hz'] = np.array(sorted(DOC_MEL_TO_HZ.items()), dtype=np.float64)
    out['doc.mel_frequencies_40'] = np.array(DOC_MEL_FREQUENCIES_40, dtype=np.float64)
    out['doc.filters_mel_22050_2048_row0'] = np.array(DOC_FILTERS_MEL_22050_2048_ROW0, dtype=np.float64)
    path = os.path.join(ROOT, 'tests', 'golden', 'mel_filterbank.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
