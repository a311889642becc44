#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE Generator.

Runs only in the build container (needs /root/reference; nothing of the reference is copied -
it is imported in place, read-only, with bytecode writing disabled).  What is committed is data:
for each case the hparams overrides and seeds that rebuild inputs and weights through
``wavthruvec_pytorch_amd.synthetic`` (so 34 MB of weights never enter the repo), the reference's
full output ``y``, small per-layer probes (first/last 32 time steps of 4 channels plus fp64 sum and
sum|.| of the whole tensor) and the post-forward buffers.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens.py
"""
import os
import sys
import types
import warnings

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference/vec2wav'
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

warnings.filterwarnings('ignore', category=FutureWarning)
import hparams as ref_hp  # noqa: E402  (reference)
import models as ref_models  # noqa: E402  (reference)

from wavthruvec_pytorch_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

CASES = [
    # name, hparams overrides, B, T, mode, extra
    dict(name='rb2_train_b2_t8', hp=dict(num_wv_feat=768), B=2, T=8, mode='train'),
    dict(name='rb2_train_b3_t17', hp=dict(num_wv_feat=768), B=3, T=17, mode='train'),
    dict(name='rb2_train_b1_t50_cfg1', hp=dict(num_wv_feat=768), B=1, T=50, mode='train'),
    dict(name='rb2_train_b2_t1', hp=dict(num_wv_feat=768), B=2, T=1, mode='train'),
    dict(name='rb1_train_b2_t8', hp=dict(num_wv_feat=768, resblock='1'), B=2, T=8, mode='train'),
    dict(name='rb2_evalcal_b2_t8', hp=dict(num_wv_feat=768), B=2, T=8, mode='evalcal'),
    dict(name='rb1_evalcal_b2_t8', hp=dict(num_wv_feat=768, resblock='1'), B=2, T=8, mode='evalcal'),
    dict(name='rb2_eval_synth_b1_t13', hp=dict(num_wv_feat=768), B=1, T=13, mode='eval'),
    dict(name='rb2_1024_x640_train_b2_t8',
         hp=dict(num_wv_feat=1024, upsample_rates=[8, 5, 4, 2, 2], upsample_kernel_sizes=[16, 11, 8, 4, 4]),
         B=2, T=8, mode='train'),
    # six upsampling stages (x640 from 512 initial channels): the last ResBlock stage has 8 channels
    dict(name='rb2_6stage_x640_train_b2_t8',
         hp=dict(num_wv_feat=768, upsample_rates=[5, 4, 4, 2, 2, 2], upsample_kernel_sizes=[11, 8, 8, 4, 4, 4]),
         B=2, T=8, mode='train'),
    dict(name='rb2_train2step_b2_t8', hp=dict(num_wv_feat=768), B=2, T=8, mode='train2'),
    dict(name='rb2_rmwn_train_b2_t8', hp=dict(num_wv_feat=768), B=2, T=8, mode='train_rmwn'),
]


def ref_hparams(overrides):
    h = types.SimpleNamespace(**{k: getattr(ref_hp, k) for k in dir(ref_hp) if not k.startswith('_') and k != 'os'})
    for k, v in overrides.items():
        setattr(h, k, v)
    return h


def probe_summary(t: torch.Tensor):
    t = t.detach()
    C, L = t.shape[1], t.shape[2]
    ch = sorted(set([0, C // 3, (2 * C) // 3, C - 1]))
    n = min(32, L)
    return dict(
        head=t[:, ch, :n].contiguous().numpy().copy(),
        tail=t[:, ch, L - n:].contiguous().numpy().copy(),
        sum=np.float64(t.double().sum().item()),
        abssum=np.float64(t.double().abs().sum().item()),
    )


def run_case(case):
    h = ref_hparams(case['hp'])
    hs = synthetic.make_hparams(**case['hp'])
    torch.manual_seed(0)
    g = ref_models.Generator(h)
    sd = synthetic.make_state_dict(hs, seed=0)
    # key list and shapes of the reference must equal the build's own spec
    ref_sd = g.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), 'state_dict key order differs from synthetic.state_dict_spec'
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    g.load_state_dict(sd)
    x, spk, noise = synthetic.make_inputs(hs, case['B'], case['T'], seed=1234)
    mode = case['mode']
    probes = {}
    hooks = []

    def add_hook(name, mod):
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: probes.__setitem__(name, probe_summary(o))))

    def hook_all():
        add_hook('conv_pre', g.conv_pre)
        for i, m in enumerate(g.ups):
            add_hook(f'ups.{i}', m)
        for i, m in enumerate(g.cbns):
            add_hook(f'cbns.{i}', m)
        for i, m in enumerate(g.resblocks):
            add_hook(f'resblocks.{i}', m)
        add_hook('conv_post', g.conv_post)

    out = {}
    with torch.no_grad():
        if mode == 'evalcal':
            for c in g.cbns:
                c.batch_nrom.momentum = 1.0
            g.train()
            g(x, spk, noise)
            g.eval()
            hook_all()
            y = g(x, spk, noise)
        elif mode == 'eval':
            g.eval()
            hook_all()
            y = g(x, spk, noise)
        elif mode == 'train2':
            g.train()
            x2, spk2, noise2 = synthetic.make_inputs(hs, case['B'], case['T'], seed=4321)
            y1 = g(x, spk, noise)
            out['y_step1'] = y1.numpy().copy()
            hook_all()
            y = g(x2, spk2, noise2)
        elif mode == 'train_rmwn':
            g.train()
            g.remove_weight_norm()
            out['keys_after_rmwn'] = np.array(list(g.state_dict().keys()))
            hook_all()
            y = g(x, spk, noise)
        else:
            g.train()
            hook_all()
            y = g(x, spk, noise)
    for hk in hooks:
        hk.remove()
    out['y'] = y.numpy().copy()
    for name, p in probes.items():
        for f, v in p.items():
            out[f'probe/{name}/{f}'] = v
    post = g.state_dict()
    for k, v in post.items():
        if 'batch_nrom' in k or k.endswith('layer.weight_u') or k.endswith('layer.weight_v'):
            out['buf/' + k] = v.numpy().copy()
    out['meta_keys'] = np.array(list(ref_sd.keys()))
    out['meta_shapes'] = np.array([','.join(map(str, ref_sd[k].shape)) for k in ref_sd])
    out['meta_case'] = np.array(repr(dict(hp=case['hp'], B=case['B'], T=case['T'], mode=mode, weight_seed=0,
                                          input_seed=1234, input_seed2=4321)))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    for case in CASES:
        out = run_case(case)
        path = os.path.join(OUT, case['name'] + '.npz')
        np.savez_compressed(path, **out)
        y = out['y']
        print(f"{case['name']:32s} y{tuple(y.shape)} |y|max={np.abs(y).max():.4f} std={y.std():.4f} "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == '__main__':
    main()
