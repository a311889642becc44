"""The streaming 32-channel stage + stride-2 upsampler (v2w_stage_bf16_n32s.hip) against fp64 math on the same bf16 operands and against the
resident-tile kernel: parity (output and BatchNorm partial sums), then interleaved timing.  (The A/B against the resident-tile kernel quoted in DESIGN.md was taken with a development switch in the dispatch, since removed: the 'old' rows of
this script now time the same kernel twice.)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wavthruvec_pytorch_amd import hipops  # noqa: E402
from tests.test_hip_ops import _wide_stage_reference  # noqa: E402

dev = torch.device('cuda:0')
C, u = 32, 2
ks, d1, d2 = [3, 7, 11], [1, 1, 1], [3, 3, 3]


def make(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, L, generator=g).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    wu = torch.randn(C, C // 2, 2 * u, generator=g) / (C * 2) ** 0.5
    bu = 0.3 * torch.randn(C // 2, generator=g)
    return x, a, s, w1, b1, w2, b2, wu, bu


def runner(x, a, s, w1, b1, w2, b2, wu, bu, with_stats=True):
    B, _, L = x.shape
    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b1=b1[j].to(dev),
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b2=b2[j].to(dev),
               k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(3)]
    wpu = hipops.pack_bf16_convt(wu.permute(2, 0, 1).contiguous().to(dev), u)
    xd, ad, sd, bud = x.to(dev), a.to(dev), s.to(dev), bu.to(dev)
    state = {}

    def call():
        nt = hipops.resblock2_stage_up_tiles(B, C, L, ks, d1, d2, slope=0.1, up_k=2 * u, up_u=u, up_slope=0.1)
        assert nt > 0
        if state.get('nt') != nt:
            state['nt'] = nt
            state['out'] = torch.full((B, C // 2, L * u), float('nan'), device=dev, dtype=torch.bfloat16)
            state['part'] = torch.full((nt * (C // 2) * 2,), float('nan'), device=dev) if with_stats else None
        ok = hipops.resblock2_stage_split(xd, (ad, sd), br, None, slope=0.1, out_div=3.0, bf16=True, io_bf16=3,
                                          up=(wpu, bud, state['out'], state['part'], 2 * u, u, 0.1))
        assert ok
        return state
    return call


def parity():
    worst = 0.0
    for B, L in [(2, 2000), (3, 4100), (1, 24), (2, 64), (2, 68), (1, 4), (2, 16384), (3, 60), (1, 1028)]:
        args = make(B, L, 300 + L)
        x, a, s, w1, b1, w2, b2, wu, bu = args
        stage, _ = _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, 0.1, True)
        z = F.leaky_relu(stage.float(), 0.1).bfloat16().double()
        want = F.conv_transpose1d(z, wu.bfloat16().double(), bu.double(), stride=u, padding=(2 * u - u) // 2)
        for off in ('', '1'):
            os.environ.pop('V2W_N32S_OFF', None)
            if off:
                os.environ['V2W_N32S_OFF'] = '1'
            st = runner(*args)()
            torch.cuda.synchronize()
            out = st['out']
            err = (out.cpu().double() - want).abs()
            bad = int((~torch.isfinite(out.float())).sum())
            sums = st['part'].view(st['nt'], C // 2, 2).double().sum(0).cpu()
            e1 = (sums[:, 0] - want.sum((0, 2))).abs().max().item()
            e2 = ((sums[:, 1] - (want * want).sum((0, 2))).abs() / (want * want).sum((0, 2))).max().item()
            tol = (err <= 2.0 ** -8 * want.abs() + 3e-2).all().item()
            print(f'B={B} L={L} {"old" if off else "new"}: nt {st["nt"]} max err {err.max().item():.3e} mean {err.mean().item():.3e} within-tol {tol} '
                  f'nonfinite {bad}  stats: sum err {e1:.3e} sumsq rel {e2:.3e}', flush=True)
            if not off:
                worst = max(worst, float('inf') if bad or not tol else err.max().item())
    os.environ.pop('V2W_N32S_OFF', None)
    return worst


def timing():
    for B, T in [(64, 512), (32, 256)]:
        L = T * 160
        call = runner(*make(B, L, 7))
        res = {}
        for rnd in range(3):
            for name, env in (('new', {}), ('old', {'V2W_N32S_OFF': '1'}), ('prio', {'V2W_STREAM_PRIO': '1'})):
                os.environ.pop('V2W_N32S_OFF', None); os.environ.pop('V2W_STREAM_PRIO', None)
                os.environ.update(env)
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    call()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f'B={B} T={T}: ' + '  '.join(f'{k} {min(v):.1f} us' for k, v in res.items()), flush=True)
    os.environ.pop('V2W_N32S_OFF', None); os.environ.pop('V2W_STREAM_PRIO', None)


if __name__ == '__main__':
    w = parity()
    print('worst new-kernel error', w)
    if w < 1.0:
        timing()
