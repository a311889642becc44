#!/usr/bin/env python3
"""Development check of v2w_branch_convs_bf16_fwd (csrc/v2w_conv_bf16_res.hip): the op against torch on the same bf16-rounded operands,
the generator with and without it (Generator.fuse_wide), and the per-launch timings of cfg3 (B=64, T=512, bf16 storage)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from wavthruvec_pytorch_amd import Generator, synthetic, hipops

dev = torch.device('cuda:0')
torch.manual_seed(0)


def ref_mode0(x, a, s, ws, bs, ks, ds, slope):
    xa = a[:, :, None] * x.float() + s[:, :, None]
    xact = F.leaky_relu(xa, slope).bfloat16().float()
    xres = torch.where(xact > 0, xact, xact / slope)
    outs = []
    for w, b, k, d in zip(ws, bs, ks, ds):
        y = F.conv1d(xact, w.bfloat16().float(), b, dilation=d, padding=d * (k - 1) // 2)
        outs.append(xres + y)
    return outs


def ref_mode1(ts, ws, bs, ks, ds, slope, div):
    tot = 0
    for t, w, b, k, d in zip(ts, ws, bs, ks, ds):
        tact = F.leaky_relu(t.float(), slope).bfloat16().float()
        tres = torch.where(tact > 0, tact, tact / slope)
        tot = tot + tres + F.conv1d(tact, w.bfloat16().float(), b, dilation=d, padding=d * (k - 1) // 2)
    return tot / div


def op_check(B, C, L):
    ks, d0, d1 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    x = torch.randn(B, C, L, device=dev).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, device=dev)
    s = 0.2 * torch.randn(B, C, device=dev)
    ws = [torch.randn(C, C, k, device=dev) / (C * k) ** 0.5 for k in ks]
    bs = [0.1 * torch.randn(C, device=dev) for _ in ks]
    wps = [hipops.pack_split(w.permute(2, 1, 0).contiguous(), bf16=True)[0] for w in ws]     # wf [k][C_in][C_out]
    t1 = [torch.empty_like(x) for _ in ks]
    ok = hipops.branch_convs_bf16(0, [x], (a, s), wps, bs, t1, ks, d0, slope=0.1)
    assert ok, 'mode 0 not served'
    want = ref_mode0(x, a, s, ws, bs, ks, d0, 0.1)
    e0 = max((g.float() - w_).abs().max().item() for g, w_ in zip(t1, want))
    r0 = max(w_.abs().max().item() for w_ in want)
    out = torch.empty_like(x)
    ok = hipops.branch_convs_bf16(1, t1, None, wps, bs, [out], ks, d1, slope=0.1, out_div=3.0)
    assert ok, 'mode 1 not served'
    want1 = ref_mode1(t1, ws, bs, ks, d1, 0.1, 3.0)
    e1 = (out.float() - want1).abs().max().item()
    print(f'op B={B} C={C} L={L}: mode0 max err {e0:.3e} (|ref| {r0:.2f})  mode1 max err {e1:.3e} (|ref| {want1.abs().max().item():.2f})')
    return e0, e1


for shape in [(2, 64, 256), (3, 64, 1000), (2, 128, 512), (3, 128, 1004), (1, 128, 20)]:
    op_check(*shape)


def stage_check(B, C, L):
    """The fused wide stage (v2w_stage_bf16_wide.hip) against torch on bf16-rounded operands (t1 kept in fp32 on the residual path)."""
    ks, d1, d2 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    x = torch.randn(B, C, L, device=dev).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, device=dev)
    s = 0.2 * torch.randn(B, C, device=dev)
    w1 = [torch.randn(C, C, k, device=dev) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, device=dev) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, device=dev) for _ in ks]
    b2 = [0.1 * torch.randn(C, device=dev) for _ in ks]
    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous(), bf16=True), b1=b1[j],
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous(), bf16=True), b2=b2[j], k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(3)]
    out = torch.empty_like(x)
    ok = hipops.resblock2_stage_split(x, (a, s), br, out, slope=0.1, out_div=3.0, bf16=True, io_bf16=3)
    assert ok, 'wide stage not served'
    xa = a[:, :, None] * x.float() + s[:, :, None]
    xact = F.leaky_relu(xa, 0.1).bfloat16().float()
    xres = torch.where(xact > 0, xact, xact / 0.1)
    tot = 0
    for j, k in enumerate(ks):
        t1 = xres + F.conv1d(xact, w1[j].bfloat16().float(), b1[j], dilation=d1[j], padding=d1[j] * (k - 1) // 2)
        tact = F.leaky_relu(t1, 0.1).bfloat16().float()
        tot = tot + t1 + F.conv1d(tact, w2[j].bfloat16().float(), b2[j], dilation=d2[j], padding=d2[j] * (k - 1) // 2)
    want = tot / 3.0
    e = (out.float() - want).abs()
    print(f'stage B={B} C={C} L={L}: max err {e.max().item():.3e} (|ref| {want.abs().max().item():.2f}) rms {e.pow(2).mean().sqrt().item():.3e} '
          f'worst at {tuple(int(v) for v in torch.nonzero(e == e.max())[0])}')


for shape in [(2, 128, 512), (3, 128, 1004), (1, 128, 20), (2, 64, 1024), (3, 64, 2000), (2, 256, 256), (3, 256, 500)]:
    stage_check(*shape)

h = synthetic.make_hparams(num_wv_feat=768)
sd = synthetic.make_state_dict(h, seed=0)


def gen(prec, fuse_wide=True):
    g = Generator(h); g.load_state_dict(sd); g = g.to(dev).train(); g.precision = prec
    g.fuse_wide = fuse_wide in (True, 'convs'); g.fuse_wide_stage = fuse_wide is True
    g.fuse_post = os.environ.get('V2W_NOPOST') is None
    return g


inp = synthetic.make_inputs(h, 4, 64, seed=3, device=dev)
with torch.no_grad():
    y32 = gen('f32')(*inp)
    yo = gen('bf16', False)(*inp)
    yn = gen('bf16', True)(*inp)
print(f'generator B=4 T=64: |bf16 old - f32| {(yo - y32).abs().max().item():.3e}  |bf16 new - f32| {(yn - y32).abs().max().item():.3e}  '
      f'rms old {(yo - y32).pow(2).mean().sqrt().item():.3e} new {(yn - y32).pow(2).mean().sqrt().item():.3e}')

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 512)
inp = synthetic.make_inputs(h, B, T, seed=4, device=dev)
for fw in (False, 'convs', True, False, 'convs', True):
    g = gen('bf16', fw)
    with torch.no_grad():
        for _ in range(3):
            g(*inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g(*inp)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        per = {}
        for _ in range(3):
            g._profile = []
            g(*inp)
            torch.cuda.synchronize()
            for tag, e0, e1 in g._profile:
                per.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3)
        g._profile = None
    print(f'--- fuse_wide={fw}: {ms:.3f} ms / forward (B={B}, T={T})')
    if per:
        for tag, ts in per.items():
            short = tag.replace('resblocks.', 'rb')
            print(f'   {short[:60]:60s} {sum(ts) / len(ts):8.1f} us')
    del g
