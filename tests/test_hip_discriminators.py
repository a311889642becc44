"""GPU parity of the MPD / MSD discriminator forwards (reference vec2wav/models.py:158-275) through the
C ABI: against the fixtures captured from the reference modules, and against the CPU oracle at other sizes."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import disc_oracle as D
from tests import golden_util
from wavthruvec_pytorch_amd import synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4      # the north_star bar on O(1) feature maps


@pytest.fixture(scope='module')
def dev():
    from wavthruvec_pytorch_amd import _hip
    _hip.load()
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def build(kind, sd, dev, training=True, precision='f32'):
    from wavthruvec_pytorch_amd.discriminators import MultiPeriodDiscriminator, MultiScaleDiscriminator, set_precision
    m = MultiPeriodDiscriminator(SimpleNamespace(periods=synthetic.DEFAULT_PERIODS)) if kind == 'mpd' else MultiScaleDiscriminator()
    m.load_state_dict(sd)
    m = set_precision(m.to(dev), precision)
    return m.train() if training else m.eval()


def _split_layers(m):
    """Layers whose last call ran on the split-f16 kernel (their weight record carries the (hi, lo) fragments)."""
    from wavthruvec_pytorch_amd.discriminators import _DiscConv
    return [l for l in m.modules() if isinstance(l, _DiscConv) and l._cache is not None and 'wps' in l._cache[1]]


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
@pytest.mark.parametrize('name', golden_util.disc_golden_names())
def test_discriminators_match_reference_goldens(dev, name, precision):
    """The fixtures captured from the reference modules, in exact fp32 and with `set_precision(m, 'f16x3')` (the dense five-tap 1024 -> 1024
    convs on the split-f16 kernel: the same 1e-4 bar)."""
    z, meta = golden_util.load_golden(name)
    sd, y, y_hat = golden_util.disc_case_setup(meta)
    m = build(meta['kind'], sd, dev, precision=precision)
    with torch.no_grad():
        if meta['mode'] == 'traineval':
            m(y_hat.to(dev), y.to(dev))
            m.eval()
        outs = m(y.to(dev), y_hat.to(dev))
    golden_util.check_disc_outputs(z, outs, TOL)
    nsplit = len(_split_layers(m))
    # f16x3: per period discriminator its four C_in > 1 convs (the three two-tap phase-stacked ones with a zero third tap), per scale
    # discriminator the grouped 512 -> 1024 / 1024 -> 1024 convs (64 output channels per group) and the dense 1024 -> 1024 one
    assert nsplit == (0 if precision == 'f32' else (12 if meta['kind'] == 'mpd' else 9)), nsplit
    for k in z.files:          # spectral-norm buffers after the forward(s): two power iterations per training forward
        if k.startswith('buf_'):
            assert np.abs(m.state_dict()[k[4:]].cpu().numpy() - z[k]).max() <= 1e-5, k


@pytest.mark.parametrize('kind,B,T', [('mpd', 3, 5120), ('msd', 3, 5120), ('mpd', 1, 247), ('msd', 1, 333), ('mpd', 2, 16000),
                                      ('msd', 2, 16001)])
def test_discriminators_match_oracle(dev, kind, B, T):
    """Every score and EVERY feature-map element against the CPU restatement (ragged lengths: not multiples of the periods,
    odd lengths through the stride-2/4 layers and the mean pools)."""
    spec = synthetic.mpd_state_dict_spec() if kind == 'mpd' else synthetic.msd_state_dict_spec()
    sd = synthetic.make_disc_state_dict(spec, seed=11)
    y, y_hat = synthetic.make_audio_pair(B, T, seed=5)
    sdo = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        want = D.mpd_forward(sdo, y, y_hat) if kind == 'mpd' else D.msd_forward(sdo, y, y_hat, training=True)
        m = build(kind, sd, dev)
        got = m(y.to(dev), y_hat.to(dev))
    for a, b in zip(want[0] + want[1], got[0] + got[1]):
        assert a.shape == b.shape and (a - b.cpu()).abs().max().item() <= TOL
    for fw, fg in zip(want[2] + want[3], got[2] + got[3]):
        assert len(fw) == len(fg)
        for a, b in zip(fw, fg):
            assert a.shape == b.shape, (a.shape, b.shape)
            assert (a - b.cpu()).abs().max().item() <= TOL


def test_discriminator_losses_and_guards(dev):
    """The loss helpers of models.py:278-310 on the HIP outputs equal the oracle's; autograd use raises (backward not built)."""
    from wavthruvec_pytorch_amd.discriminators import feature_loss, discriminator_loss, generator_loss
    sd = synthetic.make_disc_state_dict(synthetic.mpd_state_dict_spec(), seed=2)
    y, y_hat = synthetic.make_audio_pair(2, 4000, seed=9)
    with torch.no_grad():
        want = D.mpd_forward(sd, y, y_hat)
        m = build('mpd', sd, dev)
        got = m(y.to(dev), y_hat.to(dev))
        assert abs(feature_loss(got[2], got[3]).item() - feature_loss(want[2], want[3]).item()) <= 1e-4
        assert abs(discriminator_loss(got[0], got[1])[0].item() - discriminator_loss(want[0], want[1])[0].item()) <= 1e-4
        assert abs(generator_loss(got[1])[0].item() - generator_loss(want[1])[0].item()) <= 1e-4


def _hip_grads(kind, sd, y, y_hat, dev, input_grad=True, loss=None, precision='f32'):
    m = build(kind, sd, dev, precision=precision)
    yh = y_hat.to(dev).requires_grad_(input_grad)
    outs = m(y.to(dev), yh)
    (loss or D.mixed_loss)(outs).backward()
    return m, outs, yh.grad


@pytest.mark.parametrize('name', golden_util.disc_grad_golden_names())
def test_discriminator_backward_tracks_reference_gradients(dev, name):
    """loss.backward() through the HIP discriminators (train.py:188-215) against the gradients captured from the reference
    modules' own backward with the reference's losses (L1 feature loss + LSGAN terms).  Two fp32 implementations of this
    function differ by leaky_relu' / sign() flips at elements whose argument is within rounding of 0, and at this fixture's
    small B*L one flip moves upstream gradients by percents - so this is a 5 % structural check; the exact check is
    `test_discriminator_backward_matches_oracle_autograd` (masks pinned, smooth loss) with the oracle itself pinned to these
    reference gradients on the CPU at 1e-3 (tests/test_oracle_golden.py)."""
    z, meta = golden_util.load_golden(name)
    sd, y, y_hat = golden_util.disc_case_setup(meta)
    m, outs, gy = _hip_grads(meta['kind'], sd, y, y_hat, dev)
    for d in range(len(outs[0])):
        assert np.abs(outs[0][d].detach().cpu().numpy() - z[f'r{d}']).max() <= TOL
        assert np.abs(outs[1][d].detach().cpu().numpy() - z[f'g{d}']).max() <= TOL
    golden_util.check_disc_grads(z, [(k, p.grad) for k, p in m.named_parameters()], gy, rtol=5e-2, head=False)


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
@pytest.mark.parametrize('kind,B,T', [('mpd', 2, 3001), ('msd', 2, 3001), ('mpd', 1, 640), ('msd', 3, 1234)])
def test_discriminator_backward_matches_oracle_autograd(dev, kind, B, T, precision):
    """EVERY entry of EVERY gradient (all parameters incl. the spectral-normed weight_orig, and dL/dy_hat) against torch autograd
    through the CPU oracle at ragged sizes.  The oracle's leaky_relu derivative masks are pinned to the signs of the HIP
    forward's own feature maps and the loss is smooth, so the comparison is exact up to fp32 rounding."""
    spec = synthetic.mpd_state_dict_spec() if kind == 'mpd' else synthetic.msd_state_dict_spec()
    sd = synthetic.make_disc_state_dict(spec, seed=21)
    y, y_hat = synthetic.make_audio_pair(B, T, seed=6)
    m, outs, gy = _hip_grads(kind, sd, y, y_hat, dev, loss=D.smooth_loss, precision=precision)
    assert bool(_split_layers(m)) == (precision == 'f16x3')
    masks = {'r': [[t.detach().cpu() for t in fm] for fm in outs[2]], 'g': [[t.detach().cpu() for t in fm] for fm in outs[3]]}
    bufs = {k for k in sd if k.endswith('weight_u') or (k.endswith('weight_v') and k.replace('weight_v', 'weight_orig') in sd)}
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k not in bufs}
    sdo = {k: v.clone() for k, v in sd.items()}; sdo.update(leaves)
    yo = y_hat.clone().requires_grad_(True)
    want = D.mpd_forward(sdo, y, yo, masks=masks) if kind == 'mpd' else D.msd_forward(sdo, y, yo, training=True, masks=masks)
    D.smooth_loss(want).backward()
    assert (gy.cpu() - yo.grad).abs().max().item() <= 2e-4 * yo.grad.abs().max().item()
    worst = {}
    for k, p in m.named_parameters():
        ref = leaves[k].grad
        assert p.grad is not None and p.grad.shape == ref.shape, k
        worst[k] = (p.grad.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
    bad = {k: e for k, e in worst.items() if e > 5e-4}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:6]


def test_discriminator_d_step_without_input_grad(dev):
    """The D step (train.py:188-199) feeds y_g_hat.detach(): parameter gradients only, no input gradient is computed."""
    sd = synthetic.make_disc_state_dict(synthetic.mpd_state_dict_spec(), seed=2)
    y, y_hat = synthetic.make_audio_pair(2, 2000, seed=9)
    m, outs, gy = _hip_grads('mpd', sd, y, y_hat, dev, input_grad=False)
    assert gy is None and all(p.grad is not None for p in m.parameters())


def test_mpd_unfolded_tap_form(dev, monkeypatch):
    """V2W_DISC_UNFOLD=1: the k = 5 / stride 3 layers as 1-tap convs over k*C unfolded channels (v2w_unfold_taps) - same results."""
    monkeypatch.setenv('V2W_DISC_UNFOLD', '1')
    sd = synthetic.make_disc_state_dict(synthetic.mpd_state_dict_spec(), seed=4)
    y, y_hat = synthetic.make_audio_pair(2, 6000, seed=8)
    with torch.no_grad():
        want = D.mpd_forward(sd, y, y_hat)
        m = build('mpd', sd, dev)
        assert m.discriminators[0].convs[1].unfolded and not m.discriminators[0].convs[4].unfolded
        got = m(y.to(dev), y_hat.to(dev))
    for fw, fg in zip(want[2] + want[3], got[2] + got[3]):
        for a, b in zip(fw, fg):
            assert a.shape == b.shape and (a - b.cpu()).abs().max().item() <= TOL


@pytest.mark.parametrize('kind', ['mpd', 'msd'])
def test_discriminator_weights_follow_the_optimizer(dev, kind):
    """The folded / packed kernel weights are cached per parameter version: after `optimizer.step()` (train.py:199) the next forward
    must run on the updated parameters - it equals, bit for bit, a fresh module loaded from the stepped state_dict, differs from the
    forward before the step, and a second backward accumulates into .grad like torch's."""
    spec = synthetic.mpd_state_dict_spec() if kind == 'mpd' else synthetic.msd_state_dict_spec()
    sd = synthetic.make_disc_state_dict(spec, seed=13)
    y, y_hat = synthetic.make_audio_pair(2, 2500, seed=3)
    y, y_hat = y.to(dev), y_hat.to(dev)
    m = build(kind, sd, dev)
    opt = torch.optim.AdamW(m.parameters(), 1e-2, betas=(0.8, 0.99))
    out1 = m(y, y_hat)
    D.smooth_loss(out1).backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    opt.step()
    stepped = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        out2 = m(y, y_hat)
    m2 = build(kind, stepped, dev)
    with torch.no_grad():
        out3 = m2(y, y_hat)
    for a, b, c in zip(out1[1], out2[1], out3[1]):
        assert torch.equal(b, c)
        assert (a.detach() - b).abs().max().item() > 1e-4          # the step moved the scores
    for fa, fb in zip(out2[3], out3[3]):
        for a, b in zip(fa, fb):
            assert torch.equal(a, b)
    # gradients accumulate over backward calls (no zero_grad in between)
    D.smooth_loss(m(y, y_hat)).backward()
    moved = sum(int(not torch.equal(p.grad, g1[k])) for k, p in m.named_parameters())
    assert moved == len(g1)


@pytest.mark.parametrize('kind', ['mpd', 'msd'])
def test_frozen_discriminators_give_the_same_input_gradient(dev, kind):
    """`with frozen(mpd, msd):` around the generator step's discriminator forwards: dL/dy_hat is bit-identical, no parameter
    gradient is produced (train.py discards them with the next optim_d.zero_grad()), requires_grad is restored on exit."""
    from wavthruvec_pytorch_amd.discriminators import frozen
    spec = synthetic.mpd_state_dict_spec() if kind == 'mpd' else synthetic.msd_state_dict_spec()
    sd = synthetic.make_disc_state_dict(spec, seed=17)
    y, y_hat = synthetic.make_audio_pair(2, 2100, seed=12)
    m, outs, gy = _hip_grads(kind, sd, y, y_hat, dev, loss=D.smooth_loss)
    m2 = build(kind, sd, dev)
    yh = y_hat.to(dev).requires_grad_(True)
    with frozen(m2):
        assert not any(p.requires_grad for p in m2.parameters())
        outs2 = m2(y.to(dev), yh)
    D.smooth_loss(outs2).backward()
    assert all(p.requires_grad for p in m2.parameters()) and all(p.grad is None for p in m2.parameters())
    assert torch.equal(yh.grad, gy)


@pytest.mark.parametrize('kind', ['mpd', 'msd'])
def test_pair_as_one_batched_call_equals_two_calls(dev, kind, monkeypatch):
    """`BATCH_PAIRS`: (y, y_hat) of a weight-normed discriminator as ONE call on the batch [y; y_hat] - scores, every feature map, every
    parameter gradient and dL/dy_hat against the two calls of models.py:208-214 / 266-273 (a sample's result does not depend on its batch;
    only the tile choice of a launch, hence the order of its fp32 additions, may).  The spectral-normed scale discriminator keeps its two calls."""
    from wavthruvec_pytorch_amd import discriminators as HD
    spec = synthetic.mpd_state_dict_spec() if kind == 'mpd' else synthetic.msd_state_dict_spec()
    sd = synthetic.make_disc_state_dict(spec, seed=23)
    y, y_hat = synthetic.make_audio_pair(2, 2300, seed=14)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(HD, 'BATCH_PAIRS', mode)
        calls = []
        orig = HD._DiscBase.forward
        monkeypatch.setattr(HD._DiscBase, 'forward', lambda self, x, _o=orig, _c=calls: (_c.append(x.shape[0]), _o(self, x))[1])
        m = build(kind, sd, dev)
        yh = y_hat.to(dev).requires_grad_(True)
        outs = m(y.to(dev), yh)
        D.smooth_loss(outs).backward()
        monkeypatch.setattr(HD._DiscBase, 'forward', orig)
        res[mode] = (outs, {k: p.grad.clone() for k, p in m.named_parameters()}, yh.grad.clone(), calls)
    n = len(res[True][0][0])
    nsn = 0 if kind == 'mpd' else 1
    assert sorted(res[True][3]) == sorted([4] * (n - nsn) + [2] * (2 * nsn)) and res[False][3] == [2] * (2 * n)
    (sr, sg, fr, fg), (sr2, sg2, fr2, fg2) = res[True][0], res[False][0]
    for a, b in zip(sr + sg + [f for fm in fr + fg for f in fm], sr2 + sg2 + [f for fm in fr2 + fg2 for f in fm]):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    for k, g in res[True][1].items():
        g2 = res[False][1][k]
        assert (g - g2).abs().max().item() <= 2e-4 * max(g2.abs().max().item(), 1e-6), k
    assert (res[True][2] - res[False][2]).abs().max().item() <= 2e-4 * res[False][2].abs().max().item()


def _disc_ddp_worker(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda:0')
    res = {}
    for kind in ('mpd', 'msd'):
        spec = synthetic.mpd_state_dict_spec() if kind == 'mpd' else synthetic.msd_state_dict_spec()
        sd = synthetic.make_disc_state_dict(spec, seed=5)
        y, y_hat = synthetic.make_audio_pair(4, 1800, seed=40)
        lo, hi = rank * 2, rank * 2 + 2                                  # this rank's batch shard
        ys, yhs = y[lo:hi].to(dev), y_hat[lo:hi].to(dev)
        local = build(kind, sd, dev)
        D.smooth_loss(local(ys, yhs)).backward()
        ddp = DistributedDataParallel(build(kind, sd, dev))              # vec2wav/train.py:93-94
        D.smooth_loss(ddp(ys, yhs)).backward()
        res[kind] = ({k: p.grad.cpu() for k, p in local.named_parameters()},
                     {k: p.grad.cpu() for k, p in ddp.module.named_parameters()})
    torch.save(res, os.path.join(out_dir, f'disc_grad{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_ddp_wrapped_discriminators_two_ranks_one_gpu(dev, tmp_path):
    """`DistributedDataParallel(mpd)` / `(msd)` (train.py:93-94) over the HIP autograd Functions: after backward both ranks hold the
    same gradients and they are the average of the two ranks' own (un-wrapped) shard gradients.  (The discriminators have no
    cross-sample statistics, so the path needs no collective of its own: the gradient all-reduce is DDP's.)"""
    import os
    import socket
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_disc_ddp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), 'disc_grad0.pt'))
    r1 = torch.load(os.path.join(str(tmp_path), 'disc_grad1.pt'))
    for kind in ('mpd', 'msd'):
        (l0, d0), (l1, d1) = r0[kind], r1[kind]
        for k in d0:
            assert torch.equal(d0[k], d1[k]), (kind, k)
            want = (l0[k] + l1[k]) / 2
            assert (d0[k] - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1e-6), (kind, k)
