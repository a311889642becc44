"""Helpers shared by the golden-fixture tests (CPU oracle tests and GPU parity tests)."""
import ast
import glob
import os

import numpy as np
import torch

from wavthruvec_pytorch_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))


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
