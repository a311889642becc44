"""Helpers shared by the golden-fixture tests (CPU oracle tests and GPU parity tests)."""
import ast
import glob
import os

import numpy as np
import torch

from wavthruvec_pytorch_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def all_golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))


def golden_names():
    """Generator cases (the discriminator fixtures are `disc_golden_names`)."""
    return [n for n in all_golden_names() if not n.startswith(('disc_', 'mel_'))]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
    meta = ast.literal_eval(str(z['meta_case']))
    return z, meta


def case_setup(meta, device='cpu'):
    """Rebuild (h, state_dict, inputs[, inputs2]) of a golden case from its seeds."""
    h = synthetic.make_hparams(**meta['hp'])
    sd = synthetic.make_state_dict(h, seed=meta['weight_seed'], device=device)
    inp = synthetic.make_inputs(h, meta['B'], meta['T'], seed=meta['input_seed'], device=device)
    inp2 = synthetic.make_inputs(h, meta['B'], meta['T'], seed=meta['input_seed2'], device=device)
    return h, sd, inp, inp2


def probe_summary(t: torch.Tensor):
    """Same reduction tools/gen_goldens.py applied to the reference's layer outputs."""
    t = t.detach().cpu()
    C, L = t.shape[1], t.shape[2]
    ch = sorted(set([0, C // 3, (2 * C) // 3, C - 1]))
    n = min(32, L)
    return dict(head=t[:, ch, :n].numpy(), tail=t[:, ch, L - n:].numpy(),
                sum=t.double().sum().item(), abssum=t.double().abs().sum().item())


def tol_for(meta):
    """|dy| tolerance: 1e-4 is the north_star bar; the ill-conditioned synthetic-eval case gets 5e-3."""
    return 5e-3 if meta['mode'] == 'eval' else 1e-4


def fmap_probe(t: torch.Tensor):
    """Reduction applied to every discriminator feature map (B, C, L) or (B, C, H, p) by tools/gen_disc_goldens.py and by the
    tests on their own outputs: shape, head / tail of 4 channels over the flattened trailing axes, fp64 sum and sum|.|."""
    t = t.detach().cpu()
    f = t.reshape(t.shape[0], t.shape[1], -1)
    C, L = f.shape[1], f.shape[2]
    ch = sorted(set([0, C // 3, (2 * C) // 3, C - 1]))
    n = min(32, L)
    return dict(shape=np.array(t.shape, dtype=np.int64), head=f[:, ch, :n].contiguous().numpy().copy(),
                tail=f[:, ch, L - n:].contiguous().numpy().copy(),
                sum=np.float64(f.double().sum().item()), abssum=np.float64(f.double().abs().sum().item()))


def disc_golden_names():
    return [n for n in all_golden_names() if n.startswith('disc_') and not n.endswith('_grad')]


def disc_grad_golden_names():
    return [n for n in all_golden_names() if n.startswith('disc_') and n.endswith('_grad')]


def check_disc_grads(z, named_grads, grad_y_hat, rtol=2e-3, head=True):
    """Parameter gradients (through their probes: fp64 sum, sum|.|, first 16 entries) and dL/dy_hat in full against a gradient
    fixture.  Bar: rtol of the largest entry / of sum|.| of that gradient.  -> worst relative error."""
    worst = 0.0
    want = z['grad_y_hat']
    err = float(np.abs(grad_y_hat.detach().cpu().numpy() - want).max()) / float(np.abs(want).max())
    assert err <= rtol, ('grad_y_hat', err)
    worst = max(worst, err)
    seen = 0
    for k, g in named_grads:
        assert g is not None, k
        g = g.detach().cpu().reshape(-1)
        gabs = float(z['gabs_' + k])
        n = g.numel()
        scale = max(gabs / n, 1e-12)                 # mean |entry|
        e1 = abs(g.double().sum().item() - float(z['gsum_' + k])) / max(gabs, 1e-12)
        e2 = abs(g.double().abs().sum().item() - gabs) / max(gabs, 1e-12)
        e3 = float(np.abs(g[:16].numpy() - z['ghead_' + k]).max()) / max(float(np.abs(z['ghead_' + k]).max()), scale)
        # single entries: weight_v gradients are differences of nearly equal terms, and a sign() / leaky_relu' flip at a
        # borderline element moves one entry by more than rounding: 5x the bar of the sums
        assert max(e1, e2) <= rtol and (not head or e3 <= 5 * rtol), (k, e1, e2, e3)
        worst = max(worst, e1, e2, e3)
        seen += 1
    assert seen == sum(1 for f in z.files if f.startswith('gsum_')), 'parameter list differs from the fixture'
    return worst


def disc_case_setup(meta, device='cpu'):
    """(state_dict, y, y_hat) of a discriminator golden case, rebuilt from its seeds."""
    spec = synthetic.mpd_state_dict_spec() if meta['kind'] == 'mpd' else synthetic.msd_state_dict_spec()
    sd = synthetic.make_disc_state_dict(spec, seed=meta['weight_seed'], device=device)
    y, y_hat = synthetic.make_audio_pair(meta['B'], meta['T'], seed=meta['audio_seed'], device=device)
    return sd, y, y_hat


def check_disc_outputs(z, outs, tol):
    """Compare (y_d_rs, y_d_gs, fmap_rs, fmap_gs) with a golden: scores in full, fmaps through their probes.  -> worst error."""
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = outs
    worst = 0.0
    for d in range(len(y_d_rs)):
        for key, t in ((f'r{d}', y_d_rs[d]), (f'g{d}', y_d_gs[d])):
            want = z[key]
            assert tuple(t.shape) == want.shape, (key, t.shape, want.shape)
            worst = max(worst, float(np.abs(t.detach().cpu().numpy() - want).max()))
        for side, fm in (('r', fmap_rs[d]), ('g', fmap_gs[d])):
            i = 0
            while f'fmap_{side}{d}_{i}_sum' in z.files:
                pr = fmap_probe(fm[i])
                pre = f'fmap_{side}{d}_{i}_'
                assert list(pr['shape']) == list(z[pre + 'shape']), (pre, pr['shape'], z[pre + 'shape'])
                worst = max(worst, float(np.abs(pr['head'] - z[pre + 'head']).max()), float(np.abs(pr['tail'] - z[pre + 'tail']).max()))
                n = max(1, int(np.prod(pr['shape'])))
                assert abs(pr['sum'] - float(z[pre + 'sum'])) <= tol * n and abs(pr['abssum'] - float(z[pre + 'abssum'])) <= tol * n, pre
                i += 1
            assert i == len(fm), (side, d, i, len(fm))
    assert worst <= tol, worst
    return worst
