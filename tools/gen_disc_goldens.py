#!/usr/bin/env python3
"""Generate tests/golden/disc_*.npz by running the REFERENCE MultiPeriodDiscriminator / MultiScaleDiscriminator
(vec2wav/models.py:158-275), imported in place from /root/reference (build container only; nothing is copied).

Committed per case: the seeds that rebuild weights (`synthetic.make_disc_state_dict`) and audio (`synthetic.make_audio_pair`),
every score tensor in full, per-fmap probes (head / tail slices of 4 channels + fp64 sum and sum|.|) and, for the MSD, the
spectral-norm u / v buffers after the forward.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_disc_goldens.py
"""
import os
import sys
import types
import warnings

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference/vec2wav')

import numpy as np  # noqa: E402
import torch  # noqa: E402

warnings.filterwarnings('ignore', category=FutureWarning)
import hparams as ref_hp  # noqa: E402  (reference)
import models as ref_models  # noqa: E402  (reference)

from wavthruvec_pytorch_amd import synthetic  # noqa: E402
from tests.golden_util import fmap_probe  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

CASES = [
    dict(name='disc_mpd_b2_t8192', kind='mpd', B=2, T=8192, mode='train'),
    dict(name='disc_mpd_b1_t1000', kind='mpd', B=1, T=1000, mode='train'),
    dict(name='disc_msd_b2_t8192_train', kind='msd', B=2, T=8192, mode='train'),
    dict(name='disc_msd_b2_t8192_traineval', kind='msd', B=2, T=8192, mode='traineval'),   # one train forward, then eval
    dict(name='disc_msd_b1_t1000_train', kind='msd', B=1, T=1000, mode='train'),
    # backward: d(feature + generator + discriminator losses) / d(parameters, y_hat) through the reference modules
    dict(name='disc_mpd_b2_t4100_grad', kind='mpd', B=2, T=4100, mode='grad'),
    dict(name='disc_msd_b2_t4100_grad', kind='msd', B=2, T=4100, mode='grad'),
]


def mixed_loss(outs):
    """feature_loss + generator_loss + discriminator_loss of the reference (models.py:278-310) in one scalar, through the reference's
    own functions: every output of a discriminator forward gets a gradient.  (oracle/disc_oracle.py::mixed_loss is the tests' copy.)"""
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = outs
    return ref_models.feature_loss(fmap_rs, fmap_gs) + ref_models.generator_loss(y_d_gs)[0] \
        + ref_models.discriminator_loss(y_d_rs, y_d_gs)[0]


def run_case(case):
    torch.manual_seed(0)
    if case['kind'] == 'mpd':
        h = types.SimpleNamespace(periods=list(ref_hp.periods))
        assert h.periods == synthetic.DEFAULT_PERIODS
        m = ref_models.MultiPeriodDiscriminator(h)
        spec = synthetic.mpd_state_dict_spec(h.periods)
    else:
        m = ref_models.MultiScaleDiscriminator()
        spec = synthetic.msd_state_dict_spec()
    sd = synthetic.make_disc_state_dict(spec, seed=3)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), 'state_dict key order differs from the spec'
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), (k, ref_sd[k].shape, sd[k].shape)
    m.load_state_dict(sd)
    y, y_hat = synthetic.make_audio_pair(case['B'], case['T'], seed=77)
    if case['mode'] == 'grad':
        m.train()
        y_hat.requires_grad_(True)
        outs = m(y, y_hat)
        mixed_loss(outs).backward()
        out = dict(meta_case=np.array(repr(dict(case, weight_seed=3, audio_seed=77))))
        out['grad_y_hat'] = y_hat.grad.numpy().copy()
        for k, p in m.named_parameters():      # (spectral norm: parameters are bias and weight_orig)
            g = p.grad.detach().reshape(-1)
            out['gsum_' + k] = np.float64(g.double().sum().item())
            out['gabs_' + k] = np.float64(g.double().abs().sum().item())
            out['ghead_' + k] = g[:16].numpy().copy()
        for d in range(len(outs[0])):
            out[f'r{d}'] = outs[0][d].detach().numpy().copy()
            out[f'g{d}'] = outs[1][d].detach().numpy().copy()
        np.savez_compressed(os.path.join(OUT, case['name'] + '.npz'), **out)
        print(case['name'], 'max|dL/dy_hat|', y_hat.grad.abs().max().item())
        return
    with torch.no_grad():
        m.train()
        if case['mode'] == 'traineval':    # the stored u / v of a fresh state_dict are random: iterate once, then use them frozen
            m(y_hat, y)
            m.eval()
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = m(y, y_hat)
    out = dict(meta_case=np.array(repr(dict(case, weight_seed=3, audio_seed=77))))
    for d in range(len(y_d_rs)):
        out[f'r{d}'] = y_d_rs[d].numpy().copy()
        out[f'g{d}'] = y_d_gs[d].numpy().copy()
        for side, fm in (('r', fmap_rs[d]), ('g', fmap_gs[d])):
            for i, t in enumerate(fm):
                for k, v in fmap_probe(t).items():
                    out[f'fmap_{side}{d}_{i}_{k}'] = v
    if case['kind'] == 'msd':
        for k, v in m.state_dict().items():
            if k.endswith('weight_u') or (k.startswith('discriminators.0.') and k.endswith('weight_v')):
                out['buf_' + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, case['name'] + '.npz'), **out)
    print(case['name'], 'scores', [tuple(t.shape) for t in y_d_rs], 'fmaps', [tuple(t.shape) for t in fmap_rs[0]],
          'max|fmap|', max(t.abs().max().item() for t in fmap_rs[0]))


if __name__ == '__main__':
    for c in CASES:
        run_case(c)
