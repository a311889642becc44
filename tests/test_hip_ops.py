"""GPU parity of every C-ABI entry point against the oracle's arithmetic on the same seeded inputs.

All calls go through the C ABI (ctypes) of libvec2wav_hip.so.  The expected values come from the oracle
(oracle/vec2wav_oracle.py) or, for a single floating-point op, from the stock fp32 torch op it restates."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vec2wav_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    from wavthruvec_pytorch_amd import _hip
    _hip.load()
    return torch.device('cuda:0')


def _rng(seed):
    return np.random.default_rng(seed)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _relayout(w_conv):  # (Cout, Cin, k) -> [k][Cin][Cout]
    return w_conv.permute(2, 1, 0).contiguous()


@pytest.mark.parametrize('cout,cin,k', [(512, 768, 7), (256, 256, 11), (16, 16, 3), (1, 16, 7), (24, 40, 5)])
def test_wn_fold_conv(dev, cout, cin, k):
    from wavthruvec_pytorch_amd import hipops
    r = _rng(1)
    v = r.standard_normal((cout, cin, k), dtype=np.float32)
    g = (1 + 0.1 * r.standard_normal((cout, 1, 1))).astype(np.float32)
    want = _relayout(O.fold_weight_norm(torch.from_numpy(g), torch.from_numpy(v)))
    got = hipops.fold_conv_weight(_t(v, dev), _t(g, dev)).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 2e-7 * want.abs().max().item() + 1e-9
    plain = hipops.fold_conv_weight(_t(v, dev), None).cpu()       # after remove_weight_norm: relayout only
    assert torch.equal(plain, _relayout(torch.from_numpy(v)))


@pytest.mark.parametrize('cin,cout,k', [(512, 256, 11), (32, 16, 4), (512, 256, 16), (20, 12, 6)])
def test_wn_fold_convt(dev, cin, cout, k):
    from wavthruvec_pytorch_amd import hipops
    r = _rng(2)
    v = r.standard_normal((cin, cout, k), dtype=np.float32)
    g = (1 + 0.1 * r.standard_normal((cin, 1, 1))).astype(np.float32)
    w = O.fold_weight_norm(torch.from_numpy(g), torch.from_numpy(v))   # (Cin, Cout, k), norm per Cin
    want = w.permute(2, 0, 1).contiguous()                                # [k][Cin][Cout]
    got = hipops.fold_convt_weight(_t(v, dev), _t(g, dev)).cpu()
    assert (got - want).abs().max().item() <= 2e-7 * want.abs().max().item() + 1e-9


CONV_CASES = [
    # B, Cin, Cout, L, k, dil
    (2, 768, 512, 50, 7, 1),      # conv_pre, ragged L
    (2, 256, 256, 250, 3, 1),     # stage 0, L not a multiple of the tile
    (2, 256, 256, 250, 11, 3),
    (1, 128, 128, 1000, 7, 5),    # ResBlock1 dilation 5
    (2, 64, 64, 700, 11, 3),
    (3, 32, 32, 1000, 3, 1),
    (2, 16, 16, 3000, 11, 3),
    (2, 16, 16, 5, 7, 3),         # shorter than the receptive field
    (1, 16, 16, 1, 3, 1),         # single sample
    (2, 24, 40, 100, 5, 2),       # shape with no MFMA tile config -> direct kernel
    (2, 8, 8, 1999, 11, 3),       # the 8-channel stage of a six-stage generator: all output channels per thread (conv1d_small_kernel, exact width)
    (3, 16, 8, 70, 7, 1),
    (2, 6, 5, 333, 5, 2),         # ... and its guarded form (C_out < 8)
    (1, 8, 2, 9, 3, 1),
]


@pytest.mark.parametrize('algo', ['auto', 'direct'])
@pytest.mark.parametrize('B,cin,cout,L,k,dil', CONV_CASES)
def test_conv1d_fused(dev, algo, B, cin, cout, L, k, dil):
    """[affine] -> lrelu -> conv -> +bias -> +residual(affine) -> += out -> / div, all flags on."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(3)
    x = r.standard_normal((B, cin, L), dtype=np.float32)
    w = (r.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    bias = r.standard_normal(cout).astype(np.float32)
    ia, is_ = (1 + 0.2 * r.standard_normal((B, cin))).astype(np.float32), (0.3 * r.standard_normal((B, cin))).astype(np.float32)
    res = r.standard_normal((B, cout, L), dtype=np.float32)
    ra, rs = (1 + 0.2 * r.standard_normal((B, cout))).astype(np.float32), (0.3 * r.standard_normal((B, cout))).astype(np.float32)
    prev = r.standard_normal((B, cout, L), dtype=np.float32)
    tx, tw = torch.from_numpy(x), torch.from_numpy(w)
    xin = torch.from_numpy(ia)[:, :, None] * tx + torch.from_numpy(is_)[:, :, None]
    want = F.conv1d(F.leaky_relu(xin, 0.1), tw, torch.from_numpy(bias), padding=dil * (k - 1) // 2, dilation=dil)
    want = want + (torch.from_numpy(ra)[:, :, None] * torch.from_numpy(res) + torch.from_numpy(rs)[:, :, None])
    want = (want + torch.from_numpy(prev)) / 3.0
    out = _t(prev, dev).clone()
    a = hipops.ALGO_AUTO if algo == 'auto' else hipops.ALGO_DIRECT
    wf = _t(_relayout(tw).numpy(), dev)
    hipops.conv1d(_t(x, dev), wf, _t(bias, dev), out, k=k, dil=dil, slope=0.1,
                  in_affine=(_t(ia, dev), _t(is_, dev)), res=_t(res, dev), res_affine=(_t(ra, dev), _t(rs, dev)),
                  accumulate=True, out_div=3.0, algo=a, wp=hipops.pack_mfma(wf))
    err = (out.cpu() - want).abs().max().item()
    assert err <= 2e-5, f'max err {err}'


@pytest.mark.parametrize('B,cin,cout,L,k,u', [(2, 16, 8, 700, 4, 2), (3, 16, 8, 1, 4, 2), (2, 12, 5, 333, 8, 4), (1, 8, 8, 50, 4, 4), (2, 10, 3, 77, 6, 2)])
def test_convt1d_few_output_channels(dev, B, cin, cout, L, k, u):
    """leaky_relu -> ConvTranspose1d(k, stride u, padding (k - u) / 2) with C_out <= 8 (the last upsampler of a six-stage generator): one
    thread per input position, its u outputs of every channel as one vector store (convt1d_small_kernel) - against torch."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(17)
    x = torch.from_numpy(r.standard_normal((B, cin, L), dtype=np.float32))
    w = torch.from_numpy((r.standard_normal((cin, cout, k)) / np.sqrt(cin * k)).astype(np.float32))
    bias = torch.from_numpy(r.standard_normal(cout).astype(np.float32))
    want = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, bias, stride=u, padding=(k - u) // 2)
    wf = w.permute(2, 0, 1).contiguous().to(dev)          # [k][C_in][C_out]
    out = torch.full((B, cout, L * u), float('nan'), device=dev)
    hipops.convt1d(x.to(dev), wf, bias.to(dev), out, k=k, u=u, slope=0.1)
    assert (out.cpu() - want).abs().max().item() <= 2e-5


SPLIT_CASES = [
    # B, Cin, Cout, L, k, dil
    (2, 768, 512, 52, 7, 1),      # conv_pre (float4 staging: L % 4 == 0)
    (2, 256, 256, 252, 3, 1),     # 128 x 128 tiles, ragged last tile
    (2, 256, 256, 1280, 11, 3),
    (1, 128, 128, 5120, 7, 5),
    (4, 128, 128, 5120, 11, 1),   # 128 x 256 tiles
    (2, 64, 64, 700, 11, 3),      # 64 x 256 tiles
    (2, 64, 64, 701, 3, 1),       # L % 4 != 0: scalar staging
    (1, 64, 128, 9, 7, 3),        # shorter than the receptive field
]


@pytest.mark.parametrize('B,cin,cout,L,k,dil', SPLIT_CASES)
def test_conv1d_split_f16(dev, B, cin, cout, L, k, dil):
    """V2W_ALGO_SPLIT (x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on the f16 matrix pipe) against an fp64 convolution: the error must
    stay at the fp32 kernel's level (both are compared with the same fp64 reference), all epilogue flags on."""
    from wavthruvec_pytorch_amd import hipops
    assert hipops.split_supported(cin, cout)
    r = _rng(11)
    x = r.standard_normal((B, cin, L), dtype=np.float32)
    w = (r.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    w[0, 0, 0] = 1e-6                                        # a weight 6 orders below the layer maximum
    x[0, 0, : min(L, 4)] = [3e-5, -2e-7, 1e3, 0.0][: min(L, 4)]    # activations in the f16 subnormal range and a large one
    bias = r.standard_normal(cout).astype(np.float32)
    ia, is_ = (1 + 0.2 * r.standard_normal((B, cin))).astype(np.float32), (0.3 * r.standard_normal((B, cin))).astype(np.float32)
    res = r.standard_normal((B, cout, L), dtype=np.float32)
    prev = r.standard_normal((B, cout, L), dtype=np.float32)
    tx, tw = torch.from_numpy(x).double(), torch.from_numpy(w).double()
    xin = (torch.from_numpy(ia).double()[:, :, None] * tx + torch.from_numpy(is_).double()[:, :, None]).float().double()
    want = F.conv1d(F.leaky_relu(xin, 0.1), tw, torch.from_numpy(bias).double(), padding=dil * (k - 1) // 2, dilation=dil)
    want = (want + torch.from_numpy(res).double() + torch.from_numpy(prev).double()) / 3.0
    wf = _t(_relayout(torch.from_numpy(w)).numpy(), dev)
    kw = dict(k=k, dil=dil, slope=0.1, in_affine=(_t(ia, dev), _t(is_, dev)), res=_t(res, dev), accumulate=True, out_div=3.0)
    o_split = _t(prev, dev).clone()
    hipops.conv1d(_t(x, dev), None, _t(bias, dev), o_split, algo=hipops.ALGO_SPLIT, wps=hipops.pack_split(wf), **kw)
    o_f32 = _t(prev, dev).clone()
    hipops.conv1d(_t(x, dev), wf, _t(bias, dev), o_f32, algo=hipops.ALGO_MFMA, wp=hipops.pack_mfma(wf), **kw)
    e_split = (o_split.cpu().double() - want).abs().max().item()
    e_f32 = (o_f32.cpu().double() - want).abs().max().item()
    # the bar is the exact-fp32 kernel itself (both against fp64): the split products may not be worse than 2x its rounding noise
    assert e_split <= 2 * e_f32 + 1e-6, f'split max err {e_split} vs f32 kernel {e_f32}'


@pytest.mark.parametrize('B,cin,cout,L,k,dil', [SPLIT_CASES[0], SPLIT_CASES[2], SPLIT_CASES[5], SPLIT_CASES[6],
                                                (2, 32, 32, 2000, 11, 3), (3, 32, 32, 520, 3, 1), (1, 64, 32, 260, 7, 1)])      # 32 output channels (round 5)
def test_conv1d_bf16_operands(dev, B, cin, cout, L, k, dil):
    """V2W_ALGO_BF16 (BASELINE configs[2]: bf16 compute / fp32 accumulate) == an fp64 convolution of the bf16-rounded
    operands to fp32-accumulation accuracy; the rounding itself is the configuration's stated precision."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(14)
    x = r.standard_normal((B, cin, L), dtype=np.float32)
    w = (r.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    bias = r.standard_normal(cout).astype(np.float32)
    res = r.standard_normal((B, cout, L), dtype=np.float32)
    xa = F.leaky_relu(torch.from_numpy(x), 0.1).bfloat16().double()
    wb = torch.from_numpy(w).bfloat16().double()
    want = F.conv1d(xa, wb, torch.from_numpy(bias).double(), padding=dil * (k - 1) // 2, dilation=dil) + torch.from_numpy(res).double()
    wf = _t(_relayout(torch.from_numpy(w)).numpy(), dev)
    out = torch.full((B, cout, L), float('nan'), device=dev)
    hipops.conv1d(_t(x, dev), None, _t(bias, dev), out, k=k, dil=dil, slope=0.1, res=_t(res, dev), algo=hipops.ALGO_BF16,
                  wps=hipops.pack_split(wf, bf16=True))
    err = (out.cpu().double() - want).abs().max().item()
    assert err <= 2e-5, f'max err {err}'


@pytest.mark.parametrize('B,cin,L', [(4, 512, 2044), (32, 768, 256), (2, 1024, 4096)])
def test_conv_pre_bf16_one_tile_per_cu(dev, B, cin, L):
    """conv_pre of the bf16 pipeline (fp32 latents in, bf16 out: io_bf16 = 2, k = 7) on a grid of about one 128 x 128 tile per CU - the eight-wave
    instantiation conv_bf16_kernel<1, 2, 4, 2, .., 64, true> the dispatcher picks there (round 6) - against an fp64 convolution of the bf16-rounded
    operands, to the rounding of the bf16 store; ragged last tiles, sequence ends inside a tile; two launches agree bit for bit."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(77)
    x = r.standard_normal((B, cin, L), dtype=np.float32)
    w = (r.standard_normal((512, cin, 7)) / np.sqrt(cin * 7)).astype(np.float32)
    bias = r.standard_normal(512).astype(np.float32)
    want = F.conv1d(torch.from_numpy(x).bfloat16().double(), torch.from_numpy(w).bfloat16().double(), torch.from_numpy(bias).double(), padding=3)
    wf = _t(_relayout(torch.from_numpy(w)).numpy(), dev)
    wps = hipops.pack_split(wf, bf16=True)
    name = hipops.conv_bf16_config(B, 1, cin, 512, L, 7, 1, 1, io_bf16=2)
    assert name.startswith('conv_bf16_kernel<1, 2, 4, 2,'), name                 # the 128 x 128 tile on eight waves is what this shape takes
    out = torch.full((B, 512, L), float('nan'), device=dev, dtype=torch.bfloat16)
    hipops.conv1d(_t(x, dev), None, _t(bias, dev), out, k=7, dil=1, slope=1.0, algo=hipops.ALGO_BF16, wps=wps, io_bf16=2)
    assert torch.isfinite(out.float()).all()
    err = (out.cpu().double() - want).abs()
    assert (err <= 2.0 ** -8 * want.abs() + 1e-4).all(), f'max err {err.max().item()}'
    out2 = torch.full_like(out, float('nan'))
    hipops.conv1d(_t(x, dev), None, _t(bias, dev), out2, k=7, dil=1, slope=1.0, algo=hipops.ALGO_BF16, wps=wps, io_bf16=2)
    assert torch.equal(out, out2)


@pytest.mark.parametrize('B,C,L,k,dil,streams', [(3, 128, 2560, 7, 3, 'res'), (3, 128, 2560, 11, 1, 'res+add2+div'), (2, 64, 1028, 3, 1, 'acc'),
                                                  (2, 256, 300, 7, 1, 'res+add2+div'), (1, 64, 132, 3, 3, 'plain')])
def test_conv1d_bf16_activation_storage(dev, B, C, L, k, dil, streams):
    """io_bf16 = 3 (BASELINE configs[2] with bf16 activations between layers): input, residual, addends and output are bf16 TENSORS,
    arithmetic stays fp32.  Every operand-stream combination of the pipelined epilogue (residual only; residual + two addends + the
    division; running sum) against fp64 math on the same bf16 values, to the rounding of the bf16 store."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(31)
    bf = lambda a: torch.from_numpy(a).bfloat16()
    x = bf(r.standard_normal((B, C, L), dtype=np.float32))
    w = (r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    bias = r.standard_normal(C).astype(np.float32)
    ia, is_ = (1 + 0.2 * r.standard_normal((B, C))).astype(np.float32), (0.3 * r.standard_normal((B, C))).astype(np.float32)
    xin = torch.from_numpy(ia)[:, :, None].double() * x.double() + torch.from_numpy(is_)[:, :, None].double()
    xa = F.leaky_relu(xin.float(), 0.1).bfloat16().double()          # (the kernel's affine + leaky_relu run in fp32, then round)
    want = F.conv1d(xa, torch.from_numpy(w).bfloat16().double(), torch.from_numpy(bias).double(), padding=dil * (k - 1) // 2, dilation=dil)
    kw = dict(k=k, dil=dil, slope=0.1, in_affine=(_t(ia, dev), _t(is_, dev)), algo=hipops.ALGO_BF16, io_bf16=3)
    out = torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16)
    if 'res' in streams:
        res = bf(r.standard_normal((B, C, L), dtype=np.float32))
        want = want + res.double()
        kw['res'] = res.to(dev)
    if 'add2' in streams:
        a0, a1 = bf(r.standard_normal((B, C, L), dtype=np.float32)), bf(r.standard_normal((B, C, L), dtype=np.float32))
        want = want + (a0.double() + a1.double())
        kw['add'] = [a0.to(dev), a1.to(dev)]
    if 'acc' in streams:
        prev = bf(r.standard_normal((B, C, L), dtype=np.float32))
        want = want + prev.double()
        out = prev.to(dev).clone()
        kw['accumulate'] = True
    if 'div' in streams:
        want = want / 3.0
        kw['out_div'] = 3.0
    wf = _t(_relayout(torch.from_numpy(w)).numpy(), dev)
    hipops.conv1d(x.to(dev), None, _t(bias, dev), out, wps=hipops.pack_split(wf, bf16=True), **kw)
    assert out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    # the fp32 affine + activation differ from the fp64 reference by 1 bf16 ulp on a few operands; the store rounds to bf16 (2^-9 relative)
    err = (out.cpu().double() - want).abs()
    tol = 2.0 ** -8 * want.abs() + 3e-2
    assert (err <= tol).all(), f'max err {err.max().item()} (|want| max {want.abs().max().item()})'
    assert err.mean().item() <= 4e-3


@pytest.mark.parametrize('C', [32, 16])
@pytest.mark.parametrize('L', [2051, 2052, 4, 484, 7684])
def test_resblock2_stage_bf16_storage(dev, C, L):
    """The fused C = 32 / 16 stage with bf16 tensors on both sides (io_bf16 = 3) against the fp32-tensor kernel (stage_bf16_kernel, io_bf16 = 0)
    on fp32 copies of the same values.  A ragged length (L % 4 != 0) runs stage_bf16_kernel's own bf16-tensor form: the same fp32 arithmetic,
    equal up to the rounding of the bf16 store.  L % 4 == 0 runs on the resident-tile kernels (C = 32: v2w_stage_bf16_wide.hip, residual
    rebuilt from the activated bf16 operand; C = 16: v2w_stage_bf16_n16.hip, weights in registers, persistent workgroups, residual from a
    bf16 copy of x) whose t1 stays fp32 on the output path: equal to bf16 rounding.  Lengths: a row shorter than a tile (4), one valid
    window + 4 (484: the second tile holds one position quad), several tiles per row with the sequence end inside a tile."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(33)
    B = 3
    x = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32)).bfloat16().to(dev)
    a, s_ = _t((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32), dev), _t((0.3 * r.standard_normal((B, C))).astype(np.float32), dev)
    branches = []
    for k in (3, 7, 11):
        ws = [_t(_relayout(torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))).numpy(), dev) for _ in range(2)]
        branches.append(dict(wps1=hipops.pack_split(ws[0], bf16=True), b1=_t(r.standard_normal(C).astype(np.float32) * 0.1, dev),
                             wps2=hipops.pack_split(ws[1], bf16=True), b2=_t(r.standard_normal(C).astype(np.float32) * 0.1, dev), k=k, dil1=1, dil2=3))
    o16 = torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16)
    o32 = torch.full((B, C, L), float('nan'), device=dev)
    assert hipops.resblock2_stage_split(x, (a, s_), branches, o16, slope=0.1, out_div=3.0, bf16=True, io_bf16=3)
    assert hipops.resblock2_stage_split(x.float(), (a, s_), branches, o32, slope=0.1, out_div=3.0, bf16=True, io_bf16=0)
    assert torch.isfinite(o32).all()
    if L % 4:
        assert torch.equal(o16, o32.bfloat16()), f'max diff {(o16.float() - o32).abs().max().item()}'
    else:
        d = (o16.float() - o32).abs()
        assert (d <= 2.0 ** -7 * o32.abs() + 2e-2).all() and d.mean().item() <= 4e-3, f'max diff {d.max().item()}'


@pytest.mark.parametrize('B,cin,cout,L,k,dil', [(1, 768, 512, 50, 7, 1), (1, 256, 256, 250, 11, 3), (2, 128, 128, 1000, 3, 1), (1, 512, 512, 64, 7, 1)])
def test_conv1d_split_over_cin_matches_the_unsplit_kernel(dev, B, cin, cout, L, k, dil):
    """Launches of at most 128 workgroups are split over C_in chunks (slabs + splitk_reduce_kernel): same result as
    the direct kernel to fp32 summation order, bitwise identical from run to run, with every epilogue operand (residual affine,
    running sum, division), for sequence lengths that are and are not multiples of 4."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(41)
    x = _t(r.standard_normal((B, cin, L), dtype=np.float32), dev)
    wf = _t(_relayout(torch.from_numpy((r.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32))).numpy(), dev)
    bias = _t(r.standard_normal(cout).astype(np.float32), dev)
    res = _t(r.standard_normal((B, cout, L), dtype=np.float32), dev)
    ra, rs = _t((1 + 0.2 * r.standard_normal((B, cout))).astype(np.float32), dev), _t((0.3 * r.standard_normal((B, cout))).astype(np.float32), dev)
    prev = _t(r.standard_normal((B, cout, L), dtype=np.float32), dev)
    kw = dict(k=k, dil=dil, slope=0.1, res=res, res_affine=(ra, rs), accumulate=True, out_div=3.0)
    outs = []
    slab = hipops.SplitKSlab()       # caller-owned scratch (ABI v28: the library allocates nothing)
    for algo, wp, ws in ((hipops.ALGO_AUTO, hipops.pack_mfma(wf), slab), (hipops.ALGO_AUTO, hipops.pack_mfma(wf), slab),
                         (hipops.ALGO_DIRECT, None, None), (hipops.ALGO_AUTO, hipops.pack_mfma(wf), None)):
        o = prev.clone()
        hipops.conv1d(x, wf, bias, o, algo=algo, wp=wp, splitk_ws=ws, **kw)
        outs.append(o)
    # split when the launch cannot fill the chip AND its serial chain is long enough to pay for the reduce launch (>= 24 (chunk, tap) steps)
    expect_split = (cin // 32) * k >= 24
    assert (slab.t is not None and slab.t.numel() > 0) == expect_split, 'the size query and the split rule disagree'
    assert torch.equal(outs[0], outs[1]), 'split launches are not run-to-run deterministic'
    assert (outs[0] - outs[2]).abs().max().item() <= 2e-5
    assert (outs[3] - outs[2]).abs().max().item() <= 2e-5        # no slab handed over: the launch runs unsplit
    if expect_split:   # a slab smaller than the query's answer: unsplit as well (never a partial use of it)
        small = hipops.SplitKSlab(); small.t = torch.full((16,), float('nan'), device=dev)
        a = _hip_args_conv1d(x, wf, bias, prev.clone(), wp=hipops.pack_mfma(wf), small=small.t, **kw)
        assert torch.isnan(small.t).all()
        assert (a - outs[2]).abs().max().item() <= 2e-5


def _hip_args_conv1d(x, wf, bias, out, *, wp, small, **kw):
    """One v2w_conv1d_fwd call with an explicit (too small) splitk_ws buffer."""
    import ctypes as C
    from wavthruvec_pytorch_amd import _hip, hipops
    a = _hip.Conv1dArgs()
    hipops._conv1d_args(a, x, wf, bias, out, wp=wp, **kw)
    need = _hip.load().v2w_conv1d_splitk_ws_bytes(C.byref(a), 1)
    assert need > small.numel() * 4
    a.splitk_ws, a.splitk_ws_bytes = small.data_ptr(), small.numel() * 4
    _hip.check(_hip.load().v2w_conv1d_fwd(C.byref(a), hipops._stream(x)), 'v2w_conv1d_fwd')
    return out


@pytest.mark.parametrize('B,C,L,k,dil,nprob', [(4, 128, 1280, 7, 3, 3), (3, 64, 2052, 3, 1, 2), (2, 256, 520, 11, 1, 1), (4, 32, 4096, 7, 1, 3)])
def test_conv1d_masked_launch_row_sums(dev, B, C, L, k, dil, nprob):
    """rowsum_part: a masked (input-gradient) launch also writes per-tile channel sums of what it stores - the bias gradient of the layer
    whose output gradient it produces (backward of models.py:65-70) - which bn_reduce_partials adds up: == the sum of its own output."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(51)
    dy = _t(r.standard_normal((B, C, L), dtype=np.float32), dev)
    ntile = hipops.conv_rowsum_tiles(B, nprob, C, C, L, 3)
    assert ntile > 0
    probs, parts, outs = [], [], []
    for q in range(nprob):
        wf = _t(_relayout(torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))).numpy(), dev)
        msk = _t(r.standard_normal((B, C, L), dtype=np.float32), dev)
        out = torch.full((B, C, L), float('nan'), device=dev)
        part = torch.full((ntile * C * 2,), float('nan'), device=dev)
        probs.append((dy, None, None, out, dict(k=k, dil=dil, slope=1.0, res=dy, mask=(msk, None), mask_slope=0.1, wp=hipops.pack_mfma(wf),
                                               algo=hipops.ALGO_MFMA, rowsum=part)))
        parts.append(part); outs.append(out)
    if nprob > 1:
        hipops.conv1d_multi(probs)
    else:
        hipops.conv1d(*probs[0][:4], **probs[0][4])
    for part, out in zip(parts, outs):
        st = torch.empty((2 * C + 1,), device=dev, dtype=torch.float64)
        hipops.bn_reduce_partials(part, ntile, C, B * L, st)
        want = out.double().sum((0, 2))
        assert torch.isfinite(out).all()
        assert (st[:C] - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()) + 2e-3
    from wavthruvec_pytorch_amd._hip import HipLibraryError
    with pytest.raises(HipLibraryError):          # no mask: not an input-gradient launch
        hipops.conv1d(dy, None, None, outs[0], k=k, dil=dil, wp=probs[0][4]['wp'], algo=hipops.ALGO_MFMA, rowsum=parts[0])


@pytest.mark.parametrize('B,cin,cout,L,k,u', [(2, 512, 256, 50, 11, 5), (2, 256, 128, 264, 8, 4), (3, 128, 64, 1000, 8, 4),
                                              (2, 64, 32, 2052, 4, 2), (2, 32, 16, 4100, 4, 2), (1, 64, 32, 37, 4, 2),
                                              (2, 1024, 512, 40, 16, 8)])
def test_convt1d_bf16_operands(dev, B, cin, cout, L, k, u):
    """The transposed conv of BASELINE configs[2] (bf16 operands, fp32 accumulate: v2w_convt1d_bf16_fwd, run as a Conv1d over
    UP * C_out virtual channels) == an fp64 ConvTranspose1d of the bf16-rounded operands (models.py:128-129 incl. the leaky_relu in
    front), plus the fused BatchNorm partial sums of modules.py:23 against the sums of its own output."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(21)
    x = r.standard_normal((B, cin, L), dtype=np.float32)
    w = (r.standard_normal((cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
    bias = r.standard_normal(cout).astype(np.float32)
    xa = F.leaky_relu(torch.from_numpy(x), 0.1).bfloat16().double()
    wb = torch.from_numpy(w).bfloat16().double()
    want = F.conv_transpose1d(xa, wb, torch.from_numpy(bias).double(), stride=u, padding=(k - u) // 2)
    wf = _t(np.ascontiguousarray(w.transpose(2, 0, 1)), dev)                    # [k][C_in][C_out]
    wps = hipops.pack_bf16_convt(wf, u)
    assert wps is not None
    out = torch.full((B, cout, L * u), float('nan'), device=dev)
    nt = hipops.convt_bf16_stats_tiles(_t(x, dev), out, k, u)
    assert nt > 0
    part = torch.full((nt * cout * 2,), float('nan'), device=dev)
    hipops.convt1d_bf16(_t(x, dev), wps, _t(bias, dev), out, k=k, u=u, slope=0.1, stats_part=part)
    assert want.shape == out.shape
    err = (out.cpu().double() - want).abs().max().item()
    assert err <= 2e-5, f'max err {err}'
    sums = part.view(nt, cout, 2).double().sum(0).cpu()
    o = out.cpu().double()
    assert (sums[:, 0] - o.sum((0, 2))).abs().max().item() <= 1e-3 * max(1.0, o.sum((0, 2)).abs().max().item())
    assert (sums[:, 1] - (o * o).sum((0, 2))).abs().max().item() <= 1e-4 * (o * o).sum((0, 2)).max().item()


@pytest.mark.parametrize('B,cin,cout,L,k,u', [(2, 256, 128, 264, 8, 4), (3, 128, 64, 1000, 8, 4), (2, 64, 32, 2052, 4, 2),
                                              (2, 32, 16, 4100, 4, 2), (1, 64, 32, 36, 4, 2), (2, 128, 64, 72, 16, 8),
                                              (2, 512, 256, 52, 11, 5), (2, 64, 32, 37, 4, 2),
                                              (2, 512, 256, 128, 11, 5), (3, 256, 128, 192, 11, 5), (1, 512, 256, 64, 11, 5),
                                              (2, 256, 96, 128, 11, 5), (2, 512, 256, 192, 15, 5), (5, 64, 384, 64, 5, 5), (2, 64, 96, 64, 5, 5),
                                              (2, 64, 96, 64, 11, 5), (5, 64, 96, 64, 11, 5), (1, 1024, 128, 64, 11, 5)])
def test_convt1d_bf16_activation_storage(dev, B, cin, cout, L, k, u):
    """io_bf16 = 3: the transposed conv on bf16 TENSORS (models.py:128-129 under BASELINE configs[2] with bf16 activation storage).  Strides
    2 / 4 / 8 at L % 4 == 0 run on the resident-tile kernel (v2w_convt_bf16_res.hip: stores straight from the accumulators), stride 5 at
    L % 64 == 0 on its exact-phase form (C_out in steps of 128: rows co * 5 + phase, no MFMA on padding phases) or its scratch-epilogue form
    (8 virtual phases; C_out = 96 here), stride 5 otherwise and ragged lengths on the chunked kernel: all against an fp64 ConvTranspose1d of the same bf16 operands to the rounding of the bf16
    store, and the fused BatchNorm partial sums (taken from the fp32 values before rounding) against the reference's sums.  The rows of
    `stats_part` are asked for with the same tensors and io_bf16 (the tile width follows the kernel)."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(23)
    x = torch.from_numpy(r.standard_normal((B, cin, L), dtype=np.float32)).bfloat16()
    w = (r.standard_normal((cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
    bias = r.standard_normal(cout).astype(np.float32)
    xa = F.leaky_relu(x.float(), 0.1).bfloat16().double()
    want = F.conv_transpose1d(xa, torch.from_numpy(w).bfloat16().double(), torch.from_numpy(bias).double(), stride=u, padding=(k - u) // 2)
    wps = hipops.pack_bf16_convt(_t(np.ascontiguousarray(w.transpose(2, 0, 1)), dev), u)
    assert wps is not None
    xd = x.to(dev)
    out = torch.full((B, cout, L * u), float('nan'), device=dev, dtype=torch.bfloat16)
    nt = hipops.convt_bf16_stats_tiles(xd, out, k, u, io_bf16=3)
    assert nt > 0
    part = torch.full((nt * cout * 2,), float('nan'), device=dev)
    hipops.convt1d_bf16(xd, wps, _t(bias, dev), out, k=k, u=u, slope=0.1, stats_part=part, io_bf16=3)
    assert torch.isfinite(out.float()).all() and torch.isfinite(part).all()
    err = (out.cpu().double() - want).abs()
    assert (err <= 2.0 ** -8 * want.abs() + 1e-6).all(), f'max err {err.max().item()} (|want| max {want.abs().max().item()})'
    sums = part.view(nt, cout, 2).double().sum(0).cpu()
    assert (sums[:, 0] - want.sum((0, 2))).abs().max().item() <= 1e-3 * max(1.0, want.sum((0, 2)).abs().max().item())
    assert (sums[:, 1] - (want * want).sum((0, 2))).abs().max().item() <= 1e-4 * (want * want).sum((0, 2)).max().item()
    # without the statistics (eval mode) the same values
    out2 = torch.empty_like(out)
    hipops.convt1d_bf16(xd, wps, _t(bias, dev), out2, k=k, u=u, slope=0.1, io_bf16=3)
    assert torch.equal(out2, out)


def _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, slope, rebuilt_residual):
    """ResBlock2 section of a stage on bf16-rounded operands in fp64: out = mean_j [t1_j + conv2_j(lrelu(t1_j)) + b2_j],
    t1_j = x + conv1_j(lrelu(x)) + b1_j, x = a * xr + s.  rebuilt_residual: x on the residual path is the ACTIVATED bf16 operand with the
    leaky_relu undone (what the wide kernels do), else the fp32 affine result."""
    xa = (a[:, :, None] * x.float() + s[:, :, None])
    xact = F.leaky_relu(xa, slope).bfloat16().double()
    xres = torch.where(xact > 0, xact, xact / slope) if rebuilt_residual else xa.double()
    tot, t1s = 0, []
    for j, k in enumerate(ks):
        t1 = xres + F.conv1d(xact, w1[j].bfloat16().double(), b1[j].double(), dilation=d1[j], padding=d1[j] * (k - 1) // 2)
        tact = F.leaky_relu(t1.float(), slope).bfloat16().double()
        tot = tot + t1 + F.conv1d(tact, w2[j].bfloat16().double(), b2[j].double(), dilation=d2[j], padding=d2[j] * (k - 1) // 2)
        t1s.append(t1)
    return tot / len(ks), t1s


@pytest.mark.parametrize('B,C,L', [(2, 128, 512), (3, 128, 1004), (1, 128, 20), (2, 64, 1024), (3, 64, 2000), (2, 256, 256), (3, 256, 500),
                                   (2, 32, 2000), (3, 32, 4100), (1, 32, 24)])
def test_resblock2_wide_stage_bf16_storage(dev, B, C, L):
    """The whole residual section of a stage on bf16 tensors in one kernel of the resident-tile family (v2w_stage_bf16_wide.hip through
    v2w_resblock2_stage_split_fwd; models.py:135-141 with ResBlock2.forward inlined) - the wide stages C = 64 / 128 / 256, and C = 32 / 16
    (16: one k-step per tap, 32-byte rows): windows that straddle the sequence ends, ragged tile counts, the generator's kernel sizes
    and dilations - against fp64 math on the same bf16 operands, to the rounding of the bf16 store (t1 itself stays fp32 inside the
    kernel)."""
    from wavthruvec_pytorch_amd import hipops
    g = torch.Generator().manual_seed(100 + C + L)
    ks, d1, d2 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    x = torch.randn(B, C, L, generator=g).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    want, _ = _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, 0.1, True)
    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b1=b1[j].to(dev),
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b2=b2[j].to(dev),
               k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(3)]
    out = torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16)
    ok = hipops.resblock2_stage_split(x.to(dev), (a.to(dev), s.to(dev)), br, out, slope=0.1, out_div=3.0, bf16=True, io_bf16=3)
    assert ok, 'the wide stage kernel declined a shape it is built for'
    assert torch.isfinite(out.float()).all()
    err = (out.cpu().double() - want).abs()
    assert (err <= 2.0 ** -8 * want.abs() + 2e-2).all(), f'max err {err.max().item()} (|want| max {want.abs().max().item()})'
    assert err.mean().item() <= 4e-3


@pytest.mark.parametrize('B,C,L,u', [(2, 128, 512, 4), (3, 128, 1004, 4), (1, 128, 20, 4), (2, 64, 1024, 2), (3, 64, 2000, 2), (2, 256, 256, 4),
                                     (3, 256, 500, 4), (2, 32, 2000, 2), (3, 32, 4100, 2), (1, 32, 24, 2), (1, 64, 4, 2)])
@pytest.mark.parametrize('with_stats', [True, False])
def test_resblock2_wide_stage_with_the_next_upsampler_fused(dev, B, C, L, u, with_stats):
    """resblock2_stage_split(up=...): the ResBlock2 section of a stage AND the next stage's leaky_relu -> ConvTranspose1d(2u, u) + bias
    (models.py:135-141, then 128-129 of the following loop iteration) in one kernel, the stage's output never stored.  Against fp64 math
    on the same bf16 operands - the transposed conv's operand is bf16(lrelu(stage output)), ONE rounding of the fp32 value - to the
    rounding of the bf16 store; the BatchNorm partial sums (from the fp32 values) against the sums of the reference; tiles that straddle
    the sequence ends, ragged tile counts, sequences shorter than a tile."""
    from wavthruvec_pytorch_amd import hipops
    g = torch.Generator().manual_seed(300 + C + L)
    ks, d1, d2 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    x = torch.randn(B, C, L, generator=g).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    ku, cu = 2 * u, C // 2
    wu = torch.randn(C, cu, ku, generator=g) / (C * ku / u) ** 0.5
    bu = 0.3 * torch.randn(cu, generator=g)
    stage, _ = _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, 0.1, True)
    z = F.leaky_relu(stage.float(), 0.1).bfloat16().double()
    want = F.conv_transpose1d(z, wu.bfloat16().double(), bu.double(), stride=u, padding=(ku - u) // 2)
    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b1=b1[j].to(dev),
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b2=b2[j].to(dev),
               k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(3)]
    wpu = hipops.pack_bf16_convt(wu.permute(2, 0, 1).contiguous().to(dev), u)
    assert wpu is not None
    nt = hipops.resblock2_stage_up_tiles(B, C, L, ks, d1, d2, slope=0.1, up_k=ku, up_u=u, up_slope=0.1)
    assert nt > 0, 'the library declined a fused shape it is built for'
    out = torch.full((B, cu, L * u), float('nan'), device=dev, dtype=torch.bfloat16)
    part = torch.full((nt * cu * 2,), float('nan'), device=dev) if with_stats else None
    ok = hipops.resblock2_stage_split(x.to(dev), (a.to(dev), s.to(dev)), br, None, slope=0.1, out_div=3.0, bf16=True, io_bf16=3,
                                      up=(wpu, bu.to(dev), out, part, ku, u, 0.1))
    assert ok
    assert torch.isfinite(out.float()).all(), 'positions left unwritten'
    err = (out.cpu().double() - want).abs()
    # a bf16 flip of one operand element (the fp32 stage value sits on a rounding boundary) moves an output by ~2^-9 |z| |w|
    assert (err <= 2.0 ** -8 * want.abs() + 3e-2).all(), f'max err {err.max().item()} (|want| max {want.abs().max().item()})'
    assert err.mean().item() <= 5e-3
    if with_stats:
        assert torch.isfinite(part).all()
        sums = part.view(nt, cu, 2).double().sum(0).cpu()
        n = B * L * u
        # the partial rows add up to the sums of the fp32 values behind the STORED tensor (one bf16 rounding per element: 2^-9 relative, random
        # sign) - a tighter statement than a comparison with the reference's sums, which also carries the kernels' legitimate rounding choices
        # (round 6: the 32-channel streaming kernel adds t1 back as a bf16 tensor, as the reference under autocast does)
        own = out.cpu().double()
        rms = (own * own).mean().sqrt().item()
        assert (sums[:, 0] - own.sum((0, 2))).abs().max().item() <= 2.0 ** -9 * rms * (6 * n ** 0.5 + 8) + 1e-6 * n
        assert (sums[:, 1] - (own * own).sum((0, 2))).abs().max().item() <= (2e-3 + 1.6e-2 / n ** 0.5) * (own * own).sum((0, 2)).max().item()
        assert (sums[:, 0] - want.sum((0, 2))).abs().max().item() <= 1.5e-2 * n ** 0.5 + 1e-3 * want.sum((0, 2)).abs().max().item()
        assert (sums[:, 1] - (want * want).sum((0, 2))).abs().max().item() <= 4e-3 * (want * want).sum((0, 2)).max().item()
    # shapes the fused form does not exist for are declined by the query (the caller then runs the two kernels)
    assert hipops.resblock2_stage_up_tiles(B, C, L, ks, d1, d2, slope=0.1, up_k=11, up_u=5, up_slope=0.1) == 0
    assert hipops.resblock2_stage_up_tiles(B, C, L, [3, 5, 7], d1, d2, slope=0.1, up_k=ku, up_u=u, up_slope=0.1) == 0


@pytest.mark.parametrize('ks,d1,d2', [([3, 7, 11], [1, 1, 1], [3, 3, 3]), ([5], [2], [1]), ([3, 5, 7, 9], [1, 3, 1, 4], [2, 1, 5, 1])])
@pytest.mark.parametrize('B,L,affine', [(2, 1000, True), (3, 1, True), (1, 473, False), (2, 2051, True)])
def test_resblock2_stage_small_8_channels(dev, B, L, affine, ks, d1, d2):
    """v2w_resblock2_stage_small_fwd: the residual section of an 8-channel stage (the sixth stage of a x640 generator) as one fp32 FMA kernel,
    folded weights [k][C][C] - against fp64 torch; rows shorter than the receptive field, tile ends inside a row, with and without the
    conditional-BatchNorm affine, one to four branches."""
    from wavthruvec_pytorch_amd import hipops
    C = 8
    g = torch.Generator().manual_seed(11 + L + len(ks))
    x = torch.randn(B, C, L, generator=g)
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    xa = (a[:, :, None] * x + s[:, :, None]).double() if affine else x.double()
    tot = 0
    for j, k in enumerate(ks):
        t1 = xa + F.conv1d(F.leaky_relu(xa, 0.1), w1[j].double(), b1[j].double(), dilation=d1[j], padding=d1[j] * (k - 1) // 2)
        tot = tot + t1 + F.conv1d(F.leaky_relu(t1, 0.1), w2[j].double(), b2[j].double(), dilation=d2[j], padding=d2[j] * (k - 1) // 2)
    want = tot / len(ks)
    br = [dict(wf1=w1[j].permute(2, 1, 0).contiguous().to(dev), b1=b1[j].to(dev), wf2=w2[j].permute(2, 1, 0).contiguous().to(dev), b2=b2[j].to(dev),
               k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(len(ks))]
    out = torch.full((B, C, L), float('nan'), device=dev)
    assert hipops.resblock2_stage_small(x.to(dev), (a.to(dev), s.to(dev)) if affine else None, br, out, slope=0.1, out_div=float(len(ks)))
    assert (out.cpu().double() - want).abs().max().item() <= 2e-5
    # other channel counts are declined, not mis-run
    x16 = torch.zeros(1, 16, 8, device=dev)
    assert hipops.resblock2_stage_small(x16, None, [dict(wf1=torch.zeros(3, 16, 16, device=dev), b1=None, wf2=torch.zeros(3, 16, 16, device=dev), b2=None,
                                                          k=3, dil1=1, dil2=1)], torch.empty_like(x16), slope=0.1, out_div=1.0) is False


@pytest.mark.parametrize('C', [16, 32, 64])
def test_resblock2_stage_kernels_random_lengths(dev, C):
    """Seeded random (B, L) for the one-kernel stages - persistent workgroups with fewer tiles than CUs, rows of a few positions, tile counts
    that do not divide - against fp64 math on the same bf16 operands."""
    from wavthruvec_pytorch_amd import hipops
    rng = np.random.default_rng(1000 + C)
    ks, d1, d2 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    for case in range(8):
        B = int(rng.integers(1, 4))
        L = 4 * int(rng.integers(1, 900 if C == 16 else 500))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(B, C, L, generator=g).bfloat16()
        a = 1 + 0.2 * torch.randn(B, C, generator=g)
        s = 0.2 * torch.randn(B, C, generator=g)
        w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
        w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
        b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
        b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
        # (C = 16: the residual is a bf16 copy of x itself; wider: rebuilt from the activated operand)
        want, _ = _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, 0.1, C != 16)
        packed = [hipops.pack_split(w.permute(2, 1, 0).contiguous().to(dev), bf16=True) for j in range(3) for w in (w1[j], w2[j])]
        arena = torch.cat([p_[0] for p_ in packed])
        offs = np.cumsum([0] + [p_[0].numel() for p_ in packed])
        views = [(arena[offs[i]:offs[i + 1]], packed[i][1]) for i in range(6)]
        br = [dict(wps1=views[2 * j], b1=b1[j].to(dev), wps2=views[2 * j + 1], b2=b2[j].to(dev), k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(3)]
        out = torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16)
        assert hipops.resblock2_stage_split(x.to(dev), (a.to(dev), s.to(dev)), br, out, slope=0.1, out_div=3.0, bf16=True, io_bf16=3)
        err = (out.cpu().double() - want).abs()
        assert torch.isfinite(out.float()).all(), (B, L)
        assert (err <= 2.0 ** -7 * want.abs() + 2e-2).all() and err.mean().item() <= 4e-3, (B, L, err.max().item())


@pytest.mark.parametrize('ks,d1,d2', [([3, 5], [1, 2], [2, 1]), ([3, 7, 11], [1, 1, 1], [3, 3, 5]), ([9], [3], [1]), ([5, 5, 3, 7], [1, 2, 3, 1], [1, 1, 2, 4])])
@pytest.mark.parametrize('B,C,L', [(2, 128, 516), (2, 64, 1024), (1, 256, 260), (3, 32, 2000)])
def test_resblock2_wide_stage_other_block_sets(dev, B, C, L, ks, d1, d2):
    """The same kernel family on block sets OTHER than the generator's default (3, 7, 11) x (1, 3): one to four branches, other tap counts
    and dilations on either conv - the run-time-argument form of the tap loops (the default set runs a compile-time specialisation)."""
    from wavthruvec_pytorch_amd import hipops
    g = torch.Generator().manual_seed(7 + C + L + len(ks))
    nk = len(ks)
    x = torch.randn(B, C, L, generator=g).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    want, _ = _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, 0.1, True)
    # the 2 nk fragment streams back to back in execution order (include/vec2wav_hip.h: v2w_stage_split_args)
    packed = [hipops.pack_split(w.permute(2, 1, 0).contiguous().to(dev), bf16=True) for j in range(nk) for w in (w1[j], w2[j])]
    arena = torch.cat([p[0] for p in packed])
    offs = np.cumsum([0] + [p[0].numel() for p in packed])
    views = [(arena[offs[i]:offs[i + 1]], packed[i][1]) for i in range(2 * nk)]
    br = [dict(wps1=views[2 * j], b1=b1[j].to(dev), wps2=views[2 * j + 1], b2=b2[j].to(dev), k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(nk)]
    out = torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16)
    ok = hipops.resblock2_stage_split(x.to(dev), (a.to(dev), s.to(dev)), br, out, slope=0.1, out_div=float(nk), bf16=True, io_bf16=3)
    assert ok, 'the wide stage kernel declined a shape it is built for'
    assert torch.isfinite(out.float()).all()
    err = (out.cpu().double() - want).abs()
    assert (err <= 2.0 ** -8 * want.abs() + 2e-2).all(), f'max err {err.max().item()} (|want| max {want.abs().max().item()})'
    assert err.mean().item() <= 4e-3


@pytest.mark.parametrize('B,C,L', [(2, 128, 512), (3, 64, 1000), (2, 256, 260), (2, 32, 2000), (3, 16, 4100), (1, 32, 24), (1, 128, 8),
                                   (1, 16, 24), (2, 16, 504), (2, 16, 1012)])       # (16 channels: below one window, one window exactly, a seam)
@pytest.mark.parametrize('dil', [1, 3, 5])
def test_resblock1_pairs_bf16(dev, B, C, L, dil):
    """resblock1_pairs_bf16 (v2w_stage_split_args::rb1): three independent ResBlock1 pairs - kernel sizes 3 / 7 / 11, first conv at dilation
    `dil`, second at 1, each on its OWN input tensor - in one call (one launch on the resident-tile template; at 16 channels one launch per
    branch of the weights-in-registers pair kernel, round 5), then the summing form (the last problem adds the others' results and
    divides): against fp64 math on the same bf16 operands (models.py:37-44: xt = c2(lrelu(c1(lrelu x))); x = xt + x)."""
    from wavthruvec_pytorch_amd import hipops
    g = torch.Generator().manual_seed(500 + C + L + dil)
    ks = [3, 7, 11]
    xs = [torch.randn(B, C, L, generator=g).bfloat16() for _ in ks]
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s_ = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]

    def ref(x, j, affine):
        xa = (a[:, :, None] * x.float() + s_[:, :, None]) if affine else x.float()
        xact = F.leaky_relu(xa, 0.1).bfloat16().double()
        # the resident-tile kernel rebuilds the residual from the activated operand; the 16-channel kernel keeps x itself (rounded once) beside it
        xres = xa.bfloat16().double() if C == 16 else torch.where(xact > 0, xact, xact / 0.1)
        u = F.conv1d(xact, w1[j].bfloat16().double(), b1[j].double(), dilation=dil, padding=dil * (ks[j] - 1) // 2)
        uact = F.leaky_relu(u.float(), 0.1).bfloat16().double()
        return xres + F.conv1d(uact, w2[j].bfloat16().double(), b2[j].double(), padding=(ks[j] - 1) // 2)

    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b1=b1[j].to(dev),
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b2=b2[j].to(dev),
               k=ks[j], dil1=dil, dil2=1) for j in range(3)]
    fits = not (C == 256 and dil == 5)      # 256 channels at dilation 5: 94 KB of x beside 66 KB of intermediate exceed the LDS - the generator
    assert hipops.resblock1_pairs_ok(B, C, L, ks, [dil] * 3, [1] * 3, slope=0.1) == fits      # then runs that pair conv by conv
    if not fits:
        return
    xd = [x.to(dev) for x in xs]
    for affine in (True, False):
        outs = [torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16) for _ in ks]
        ok = hipops.resblock1_pairs_bf16(xd, (a.to(dev), s_.to(dev)) if affine else None, br, outs, slope=0.1)
        assert ok
        for j in range(3):
            want = ref(xs[j], j, affine)
            err = (outs[j].cpu().double() - want).abs()
            assert torch.isfinite(outs[j].float()).all()
            assert (err <= 2.0 ** -8 * want.abs() + 2e-2).all(), f'branch {j}: max err {err.max().item()} (|want| max {want.abs().max().item()})'
    # the last pair of a stage: branches 0 and 1 first, then branch 2 takes their (bf16) results and divides
    o01 = [torch.empty((B, C, L), device=dev, dtype=torch.bfloat16) for _ in range(2)]
    assert hipops.resblock1_pairs_bf16(xd[:2], None, br[:2], o01, slope=0.1)
    tot = torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16)
    assert hipops.resblock1_pairs_bf16(xd[2:], None, br[2:], [tot], slope=0.1, out_div=3.0, add=o01)
    want = ((o01[0].cpu().double() + o01[1].cpu().double()) + ref(xs[2], 2, False)) / 3.0
    err = (tot.cpu().double() - want).abs()
    assert (err <= 2.0 ** -8 * want.abs() + 1e-2).all(), f'sum: max err {err.max().item()}'


@pytest.mark.parametrize('kp', [7, 9])
@pytest.mark.parametrize('B,L', [(2, 1000), (3, 4100), (1, 24), (2, 216), (2, 220), (2, 224), (2, 468), (2, 472), (2, 476), (1, 948)])
def test_resblock2_stage16_with_the_fused_tail(dev, B, L, kp):
    """The last (C = 16) stage with leaky_relu(0.01) -> conv_post -> tanh (models.py:143-145) inside the same kernel: the stage's output is
    not written, the fp32 audio is - against fp64 math on the same bf16 operands.  kp = 7 (the reference's conv_post): the streaming kernel
    (v2w_stage_bf16_n16s.hip: one wave per workgroup walks runs of the sequence, run seams at multiples of 64 positions); kp = 9: the
    resident-tile template (220 outputs per tile)."""
    from wavthruvec_pytorch_amd import hipops
    C = 16
    g = torch.Generator().manual_seed(300 + L)
    ks, d1, d2 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    x = torch.randn(B, C, L, generator=g).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    wpost = torch.randn(1, C, kp, generator=g) / (C * kp) ** 0.5
    bpost = 0.1 * torch.randn(1, generator=g)
    out_want, _ = _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, 0.1, True)
    y_want = torch.tanh(F.conv1d(F.leaky_relu(out_want, 0.01), wpost.double(), bpost.double(), padding=(kp - 1) // 2))
    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b1=b1[j].to(dev),
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b2=b2[j].to(dev),
               k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(3)]
    y = torch.full((B, 1, L), float('nan'), device=dev)
    ok = hipops.resblock2_stage_split(x.to(dev), (a.to(dev), s.to(dev)), br, None, slope=0.1, out_div=3.0, bf16=True, io_bf16=3,
                                      post=(wpost.permute(2, 1, 0).contiguous().to(dev), bpost.to(dev), y, kp, 0.01))
    assert ok, 'the fused tail was declined'
    assert torch.isfinite(y).all()
    err = (y.cpu().double() - y_want).abs()
    # kp = 7 (round 6: the streaming kernel, v2w_stage_bf16_n16s.hip): conv_post runs on the matrix pipe, its operand lrelu(stage output) is rounded
    # to bf16 ONCE - the arithmetic of every other conv of this mode and of the reference under autocast - against weights kept to 16 mantissa
    # bits (hi + lo rows): with these unit-scale random tail weights that is ~7e-4 mean, < 1e-2 max (the two-kernel form below differs as much)
    bound, mean_bound = (1e-2, 1.5e-3) if kp == 7 else (5e-3, 5e-4)
    assert err.max().item() <= bound, f'max err {err.max().item()} at {tuple(int(v) for v in torch.nonzero(err == err.max())[0])}'
    assert err.mean().item() <= mean_bound, err.mean().item()
    # == the two-kernel form (stage output rounded to bf16, then v2w_conv_post_tanh_bf16in) to the rounding of that tensor
    out = torch.empty((B, C, L), device=dev, dtype=torch.bfloat16)
    assert hipops.resblock2_stage_split(x.to(dev), (a.to(dev), s.to(dev)), br, out, slope=0.1, out_div=3.0, bf16=True, io_bf16=3)
    y2 = torch.empty_like(y)
    hipops.conv_post_tanh(out, wpost.permute(2, 1, 0).contiguous().to(dev), bpost.to(dev), y2, k=kp, slope=0.01)
    assert (y2 - y).abs().max().item() <= 2e-2


@pytest.mark.parametrize('C', [16, 32])
@pytest.mark.parametrize('affine,biases,out_div', [(False, True, 3.0), (True, False, 3.0), (True, True, 0.0), (False, False, 0.0)])
def test_streaming_narrow_stages_optional_arguments(dev, C, affine, biases, out_div):
    """The round-6 streaming kernels (v2w_stage_bf16_n16s.hip: 16 channels + the 7-tap tail; v2w_stage_bf16_n32s.hip: 32 channels + the
    stride-2 upsampler) with every optional argument of v2w_stage_split_args absent in turn: no CondBN affine (in_a / in_s NULL), no biases
    (conv, tail and upsampler), no division of the branch sum (out_div = 0) - against fp64 math on the same bf16 operands.  Run seams: L spans
    several runs of every wave / team and ends inside a block."""
    from wavthruvec_pytorch_amd import hipops
    B, L = 3, 2 * 4096 + 52
    g = torch.Generator().manual_seed(900 + C + int(affine) + 2 * int(biases))
    ks, d1, d2 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    x = torch.randn(B, C, L, generator=g).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, generator=g) if affine else torch.ones(B, C)
    s = 0.2 * torch.randn(B, C, generator=g) if affine else torch.zeros(B, C)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    zero = torch.zeros(C)
    b1 = [0.1 * torch.randn(C, generator=g) if biases else zero for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) if biases else zero for _ in ks]
    stage3, _ = _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, 0.1, C != 16)      # (= sum / 3)
    stage = stage3 if out_div != 0.0 else stage3 * 3.0
    br = [dict(wps1=hipops.pack_split(w1[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b1=b1[j].to(dev) if biases else None,
               wps2=hipops.pack_split(w2[j].permute(2, 1, 0).contiguous().to(dev), bf16=True), b2=b2[j].to(dev) if biases else None,
               k=ks[j], dil1=d1[j], dil2=d2[j]) for j in range(3)]
    aff = (a.to(dev), s.to(dev)) if affine else None
    if C == 16:
        wpost = torch.randn(1, C, 7, generator=g) / (C * 7) ** 0.5 / (1.0 if out_div != 0.0 else 3.0)
        bpost = 0.1 * torch.randn(1, generator=g) if biases else None
        want = torch.tanh(F.conv1d(F.leaky_relu(stage, 0.01), wpost.double(), None if bpost is None else bpost.double(), padding=3))
        y = torch.full((B, 1, L), float('nan'), device=dev)
        ok = hipops.resblock2_stage_split(x.to(dev), aff, br, None, slope=0.1, out_div=out_div, bf16=True, io_bf16=3,
                                          post=(wpost.permute(2, 1, 0).contiguous().to(dev), None if bpost is None else bpost.to(dev), y, 7, 0.01))
        assert ok and torch.isfinite(y).all()
        err = (y.cpu().double() - want).abs()
        assert err.max().item() <= 1.5e-2 and err.mean().item() <= 2e-3, (err.max().item(), err.mean().item())
    else:
        u = 2
        wu = torch.randn(C, C // 2, 2 * u, generator=g) / (C * 2) ** 0.5 / (1.0 if out_div != 0.0 else 3.0)
        bu = 0.3 * torch.randn(C // 2, generator=g) if biases else None
        z = F.leaky_relu(stage.float(), 0.1).bfloat16().double()
        want = F.conv_transpose1d(z, wu.bfloat16().double(), None if bu is None else bu.double(), stride=u, padding=u // 2)
        wpu = hipops.pack_bf16_convt(wu.permute(2, 0, 1).contiguous().to(dev), u)
        nt = hipops.resblock2_stage_up_tiles(B, C, L, ks, d1, d2, slope=0.1, up_k=2 * u, up_u=u, up_slope=0.1)
        assert nt > 0
        out = torch.full((B, C // 2, L * u), float('nan'), device=dev, dtype=torch.bfloat16)
        part = torch.full((nt * (C // 2) * 2,), float('nan'), device=dev)
        ok = hipops.resblock2_stage_split(x.to(dev), aff, br, None, slope=0.1, out_div=out_div, bf16=True, io_bf16=3,
                                          up=(wpu, None if bu is None else bu.to(dev), out, part, 2 * u, u, 0.1))
        assert ok and torch.isfinite(out.float()).all() and torch.isfinite(part).all()
        err = (out.cpu().double() - want).abs()
        assert (err <= 2.0 ** -8 * want.abs() + 4e-2).all() and err.mean().item() <= 6e-3, (err.max().item(), err.mean().item())
        own = out.cpu().double()
        sums = part.view(nt, C // 2, 2).double().sum(0).cpu()
        n = B * L * u
        rms = (own * own).mean().sqrt().item()
        assert (sums[:, 0] - own.sum((0, 2))).abs().max().item() <= 2.0 ** -9 * rms * (6 * n ** 0.5 + 8) + 1e-6 * n
        assert (sums[:, 1] - (own * own).sum((0, 2))).abs().max().item() <= (2e-3 + 1.6e-2 / n ** 0.5) * (own * own).sum((0, 2)).max().item()


@pytest.mark.parametrize('B,C,L', [(2, 64, 256), (3, 64, 1000), (2, 128, 512), (3, 128, 1004), (1, 128, 20)])
def test_branch_convs_bf16(dev, B, C, L):
    """v2w_branch_convs_bf16_fwd: the first convs of all branches of a wide stage in one launch (mode 0: x staged once, three t1
    tensors), then the second convs on one accumulator (mode 1: one output) - the two-launch form of a wide stage, against fp64 math
    on the same bf16 operands (the residuals rebuilt from the activated operand, as the kernel does)."""
    from wavthruvec_pytorch_amd import hipops
    g = torch.Generator().manual_seed(7 + C + L)
    ks, d1, d2 = [3, 7, 11], [1, 1, 1], [3, 3, 3]
    x = torch.randn(B, C, L, generator=g).bfloat16()
    a = 1 + 0.2 * torch.randn(B, C, generator=g)
    s = 0.2 * torch.randn(B, C, generator=g)
    w1 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    w2 = [torch.randn(C, C, k, generator=g) / (C * k) ** 0.5 for k in ks]
    b1 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    b2 = [0.1 * torch.randn(C, generator=g) for _ in ks]
    _, t1_want = _wide_stage_reference(x, a, s, w1, b1, w2, b2, ks, d1, d2, 0.1, True)
    wp1 = [hipops.pack_split(w.permute(2, 1, 0).contiguous().to(dev), bf16=True)[0] for w in w1]
    wp2 = [hipops.pack_split(w.permute(2, 1, 0).contiguous().to(dev), bf16=True)[0] for w in w2]
    t1 = [torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16) for _ in ks]
    assert hipops.branch_convs_bf16(0, [x.to(dev)], (a.to(dev), s.to(dev)), wp1, [b.to(dev) for b in b1], t1, ks, d1, slope=0.1)
    for got, want in zip(t1, t1_want):
        err = (got.cpu().double() - want).abs()
        assert (err <= 2.0 ** -8 * want.abs() + 1e-2).all(), f'mode 0: max err {err.max().item()}'
    # mode 1 on the kernel's own (bf16) t1 tensors
    tot = 0
    for j, k in enumerate(ks):
        tact = F.leaky_relu(t1[j].cpu().float(), 0.1).bfloat16().double()
        tres = torch.where(tact > 0, tact, tact / 0.1)
        tot = tot + tres + F.conv1d(tact, w2[j].bfloat16().double(), b2[j].double(), dilation=d2[j], padding=d2[j] * (k - 1) // 2)
    want = tot / 3.0
    out = torch.full((B, C, L), float('nan'), device=dev, dtype=torch.bfloat16)
    assert hipops.branch_convs_bf16(1, t1, None, wp2, [b.to(dev) for b in b2], [out], ks, d2, slope=0.1, out_div=3.0)
    err = (out.cpu().double() - want).abs()
    assert (err <= 2.0 ** -8 * want.abs() + 1e-2).all(), f'mode 1: max err {err.max().item()}'


@pytest.mark.parametrize('C', [32, 16])
@pytest.mark.parametrize('B,L,bf16', [(2, 1000, False), (3, 4099, False), (1, 300, False), (2, 1000, True)])
def test_resblock2_stage_split(dev, B, L, bf16, C):
    """Fused C = 32 / 16 ResBlock2 stage on the f16 (split) / bf16 matrix pipe against the fp64 math of models.py:65-70,135-141."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(15)
    nk = 3
    x = r.standard_normal((B, C, L), dtype=np.float32)
    a, s_ = (1 + 0.2 * r.standard_normal((B, C))).astype(np.float32), (0.3 * r.standard_normal((B, C))).astype(np.float32)
    xin = (torch.from_numpy(a)[:, :, None] * torch.from_numpy(x) + torch.from_numpy(s_)[:, :, None]).double()
    branches, want = [], 0
    ks = (3, 7, 11)
    arena = torch.zeros((sum(2 * hipops.split_units_halves(k, C, C) for k in ks) + 1024,), device=dev, dtype=torch.float16)
    off = 0                                                      # the kernel wants the six streams back to back
    for k in ks:
        ws = [(r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32) for _ in range(2)]
        bs = [r.standard_normal(C).astype(np.float32) * 0.1 for _ in range(2)]
        q = lambda t: t.bfloat16().double() if bf16 else t.double()      # the configuration's stated operand precision
        t1 = xin + F.conv1d(q(F.leaky_relu(xin, 0.1)), q(torch.from_numpy(ws[0])), torch.from_numpy(bs[0]).double(), padding=(k - 1) // 2)
        rj = t1 + F.conv1d(q(F.leaky_relu(t1, 0.1)), q(torch.from_numpy(ws[1])), torch.from_numpy(bs[1]).double(), padding=3 * (k - 1) // 2, dilation=3)
        want = want + rj
        wfs = [_t(_relayout(torch.from_numpy(w)).numpy(), dev) for w in ws]
        n = hipops.split_units_halves(k, C, C)
        packed = [hipops.pack_split(wfs[i], out=arena[off + i * n: off + (i + 1) * n], bf16=bf16) for i in range(2)]
        off += 2 * n
        branches.append(dict(wps1=packed[0], b1=_t(bs[0], dev), wps2=packed[1], b2=_t(bs[1], dev), k=k, dil1=1, dil2=3))
    want = want / nk
    out = torch.full((B, C, L), float('nan'), device=dev)
    assert hipops.resblock2_stage_split(_t(x, dev), (_t(a, dev), _t(s_, dev)), branches, out, slope=0.1, out_div=float(nk), bf16=bf16)
    err = (out.cpu().double() - want).abs().max().item()
    assert err <= (2e-2 if bf16 else 2e-5), f'max err {err}'    # bf16: t1 itself is only carried with 8 bits through the LDS tile
    from wavthruvec_pytorch_amd._hip import HipLibraryError
    branches[1], branches[2] = branches[2], branches[1]          # streams no longer back to back in execution order
    if bf16:     # the bf16 stage kernel (v2w_stage_bf16.hip) addresses every stream by its own pointer: any placement is fine
        assert hipops.resblock2_stage_split(_t(x, dev), (_t(a, dev), _t(s_, dev)), branches, out, slope=0.1, out_div=float(nk), bf16=True)
    else:
        with pytest.raises(HipLibraryError):
            hipops.resblock2_stage_split(_t(x, dev), (_t(a, dev), _t(s_, dev)), branches, out, slope=0.1, out_div=float(nk), bf16=bf16)


def test_conv1d_split_multi_and_rejects(dev):
    from wavthruvec_pytorch_amd import hipops
    assert not hipops.split_supported(16, 16) and not hipops.split_supported(32, 32) and not hipops.split_supported(64, 64, 2)
    r = _rng(12)
    B, C, L = 2, 128, 640
    x = _t(r.standard_normal((B, C, L), dtype=np.float32), dev)
    probs, wants = [], []
    for k, d in ((11, 1), (7, 3), (3, 1)):
        w = (r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
        wf = _t(_relayout(torch.from_numpy(w)).numpy(), dev)
        out = torch.full((B, C, L), float('nan'), device=dev)
        probs.append((x, None, None, out, dict(k=k, dil=d, slope=0.1, res=x, algo=hipops.ALGO_SPLIT, wps=hipops.pack_split(wf))))
        wants.append(x.cpu() + F.conv1d(F.leaky_relu(x.cpu(), 0.1), torch.from_numpy(w), None, padding=d * (k - 1) // 2, dilation=d))
    hipops.conv1d_multi(probs)
    for (_, _, _, out, _), want in zip(probs, wants):
        assert (out.cpu() - want).abs().max().item() <= 2e-5
    from wavthruvec_pytorch_amd._hip import HipLibraryError
    with pytest.raises(HipLibraryError):          # missing fragments
        hipops.conv1d(x, None, None, probs[0][3], k=3, algo=hipops.ALGO_SPLIT)


def test_split_pack_batch_equals_per_layer(dev):
    """v2w_split_pack_batch (weight-norm fold + split pack of many layers, three launches) == fold_conv_weight + pack_split."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(13)
    layers, want = [], []
    for cout, cin, k, wn in ((512, 768, 7, True), (256, 256, 11, True), (128, 128, 3, True), (64, 64, 7, False)):
        v = _t(r.standard_normal((cout, cin, k), dtype=np.float32), dev)
        g = _t((1 + 0.1 * r.standard_normal((cout, 1, 1))).astype(np.float32), dev) if wn else None
        wps = torch.zeros((hipops.split_halves(k, cin, cout),), device=dev, dtype=torch.float16)
        sc = torch.zeros((4,), device=dev)
        layers.append((v, g, wps, sc))
        want.append(hipops.pack_split(hipops.fold_conv_weight(v, g)))
    hipops.SplitPlan(layers, dev).run()
    for (v, g, wps, sc), (wps0, sc0) in zip(layers, want):
        n = wps.numel() - 1024                                   # the tail unit is padding
        assert torch.equal(sc[:2].cpu(), sc0[:2].cpu())
        assert torch.equal(wps[:n].view(torch.int16).cpu(), wps0[:n].view(torch.int16).cpu())


@pytest.mark.parametrize('B,cin,cout,L,k,dil', CONV_CASES[:7])
def test_conv1d_plain_and_mfma_forced(dev, B, cin, cout, L, k, dil):
    """No optional inputs; V2W_ALGO_MFMA must accept every generator shape and agree with the direct kernel."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(4)
    x = r.standard_normal((B, cin, L), dtype=np.float32)
    w = (r.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    want = F.conv1d(torch.from_numpy(x), torch.from_numpy(w), None, padding=dil * (k - 1) // 2, dilation=dil)
    wf = _t(_relayout(torch.from_numpy(w)).numpy(), dev)
    o1 = torch.full((B, cout, L), float('nan'), device=dev)
    o2 = torch.full((B, cout, L), float('nan'), device=dev)
    wp = hipops.pack_mfma(wf)
    assert wp is not None
    hipops.conv1d(_t(x, dev), None, None, o1, k=k, dil=dil, slope=1.0, algo=hipops.ALGO_MFMA, wp=wp)
    hipops.conv1d(_t(x, dev), wf, None, o2, k=k, dil=dil, slope=1.0, algo=hipops.ALGO_DIRECT)
    assert (o1.cpu() - want).abs().max().item() <= 2e-5
    assert (o2.cpu() - want).abs().max().item() <= 2e-5


def test_conv1d_rejects_bad_arguments(dev):
    from wavthruvec_pytorch_amd import hipops, _hip
    x = torch.zeros((1, 24, 10), device=dev)
    wf = torch.zeros((5, 24, 40), device=dev)
    out = torch.zeros((1, 40, 10), device=dev)
    assert hipops.pack_mfma(wf) is None                                 # no MFMA tile config for 24 -> 40
    with pytest.raises(_hip.HipLibraryError):
        hipops.conv1d(x, wf, None, out, k=5, algo=hipops.ALGO_MFMA)
    with pytest.raises(_hip.HipLibraryError):
        hipops.conv1d(x, wf, None, out, k=4)                            # even kernel size
    with pytest.raises(_hip.HipLibraryError):
        hipops.conv1d(x, wf, None, out, k=5, algo=7)


CONVT_CASES = [
    # B, Cin, Cout, L, k, u
    (2, 512, 256, 50, 11, 5),
    (2, 256, 128, 250, 8, 4),
    (2, 128, 64, 333, 8, 4),
    (2, 64, 32, 1001, 4, 2),
    (2, 32, 16, 2000, 4, 2),
    (2, 512, 256, 17, 16, 8),     # x640 variant, first stage
    (2, 256, 128, 136, 11, 5),    # x640 variant, second stage
    (1, 512, 256, 1, 11, 5),      # single frame
    (2, 20, 12, 40, 6, 2),        # no tile config -> direct
    (2, 64, 32, 100, 9, 3),       # stride without a tile instantiation -> direct
]


@pytest.mark.parametrize('algo', ['auto', 'direct'])
@pytest.mark.parametrize('B,cin,cout,L,k,u', CONVT_CASES)
def test_convt1d(dev, algo, B, cin, cout, L, k, u):
    from wavthruvec_pytorch_amd import hipops
    r = _rng(5)
    x = r.standard_normal((B, cin, L), dtype=np.float32)
    w = (r.standard_normal((cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
    bias = r.standard_normal(cout).astype(np.float32)
    want = F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x), 0.1), torch.from_numpy(w), torch.from_numpy(bias),
                              stride=u, padding=(k - u) // 2)
    assert want.shape[2] == L * u
    wf = _t(torch.from_numpy(w).permute(2, 0, 1).contiguous().numpy(), dev)
    out = torch.full((B, cout, L * u), float('nan'), device=dev)
    a = hipops.ALGO_AUTO if algo == 'auto' else hipops.ALGO_DIRECT
    nt = hipops.convt_stats_tiles(B, cin, cout, L, k, u) if algo == 'auto' else 0
    part = torch.full((nt * cout * 2,), float('nan'), device=dev) if nt else None
    hipops.convt1d(_t(x, dev), wf, _t(bias, dev), out, k=k, u=u, slope=0.1, algo=a, wp=hipops.pack_mfma(wf, u=u),
                   stats_part=part)
    err = (out.cpu() - want).abs().max().item()
    assert err <= 2e-5, f'max err {err}'
    if nt:   # fused BatchNorm statistics of the output: per-tile partials -> fixed-order fp64 reduction
        stats = torch.empty((2 * cout + 1,), device=dev, dtype=torch.float64)
        hipops.bn_reduce_partials(part, nt, cout, B * L * u, stats)
        st = stats.cpu()
        s1 = want.double().sum(dim=(0, 2)); s2 = want.double().pow(2).sum(dim=(0, 2))
        assert (st[:cout] - s1).abs().max().item() <= 1e-5 * max(1.0, s2.max().item() ** 0.5 * (B * L * u) ** 0.5)
        assert ((st[cout:2 * cout] - s2).abs() / s2).max().item() <= 1e-5
        assert st[2 * cout].item() == B * L * u


@pytest.mark.parametrize('training', [True, False])
def test_cond_gamma_beta(dev, training):
    """fcs[i] + spectral-norm power iteration + Linear for all five stages in one call; u/v mutate in train only."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(6)
    B = 5
    Cs = [256, 128, 64, 32, 16]
    spk = r.standard_normal((B, 192), dtype=np.float32)
    nz = r.standard_normal((B, 192), dtype=np.float32)
    spk_noise = torch.from_numpy(np.concatenate([spk, nz], 1))
    fc_w, fc_b, sn_w, sn_b, sn_u, sn_v, gb, want_gb, want_u, want_v = [], [], [], [], [], [], [], [], [], []
    for C in Cs:
        fw = (r.standard_normal((128, 384)) / np.sqrt(384)).astype(np.float32)
        fb = (0.1 * r.standard_normal(128)).astype(np.float32)
        w = (1 + 0.02 * r.standard_normal((2 * C, 128))).astype(np.float32)
        b = (0.1 * r.standard_normal(2 * C)).astype(np.float32)
        u = r.standard_normal(2 * C); u = (u / np.linalg.norm(u)).astype(np.float32)
        v = r.standard_normal(128); v = (v / np.linalg.norm(v)).astype(np.float32)
        z = F.linear(spk_noise, torch.from_numpy(fw), torch.from_numpy(fb))
        w_sn, u2, v2 = O.spectral_norm_weight(torch.from_numpy(w), torch.from_numpy(u), torch.from_numpy(v), training)
        want_gb.append(F.linear(z, w_sn, torch.from_numpy(b)))
        want_u.append(u2); want_v.append(v2)
        fc_w.append(_t(fw, dev)); fc_b.append(_t(fb, dev)); sn_w.append(_t(w, dev)); sn_b.append(_t(b, dev))
        sn_u.append(_t(u, dev)); sn_v.append(_t(v, dev)); gb.append(torch.empty((B, 2 * C), device=dev))
    z_ws = torch.empty((5 * B * 128,), device=dev)
    sg = torch.empty((5,), device=dev)
    hipops.cond_gamma_beta(_t(spk, dev), _t(nz, dev), fc_w, fc_b, sn_w, sn_b, sn_u, sn_v, gb, z_ws, sg, training)
    for i in range(5):
        scale = want_gb[i].abs().max().item()
        assert (gb[i].cpu() - want_gb[i]).abs().max().item() <= 2e-5 * max(1.0, scale)
        assert (sn_u[i].cpu() - want_u[i]).abs().max().item() <= 1e-6
        assert (sn_v[i].cpu() - want_v[i]).abs().max().item() <= 1e-6


@pytest.mark.parametrize('B', [1, 5])
def test_cond_affine_eval_one_launch(dev, B):
    """Eval mode: v2w_cond_sigma (once per weight version) + v2w_cond_affine_eval (ONE launch per forward, every stage) == the oracle's
    fcs -> spectral-norm Linear -> eval-mode BatchNorm affine folded into (a, s) (models.py:120,131-133, modules.py:20-30), and == the
    three-launch v2w_cond_gamma_beta + five v2w_bn_finalize(training = 0) path it replaces; u / v untouched."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(61)
    Cs = [256, 128, 64, 32, 16, 8]
    spk = r.standard_normal((B, 192), dtype=np.float32)
    nz = r.standard_normal((B, 192), dtype=np.float32)
    spk_noise = torch.from_numpy(np.concatenate([spk, nz], 1)).double()
    fc_w, fc_b, sn_w, sn_b, sn_u, sn_v, rms, rvs, want_a, want_s = [], [], [], [], [], [], [], [], [], []
    for C in Cs:
        fw = (r.standard_normal((128, 384)) / np.sqrt(384)).astype(np.float32)
        fb = (0.1 * r.standard_normal(128)).astype(np.float32)
        w = (1 + 0.02 * r.standard_normal((2 * C, 128))).astype(np.float32)
        b = (0.1 * r.standard_normal(2 * C)).astype(np.float32)
        u = r.standard_normal(2 * C); u = (u / np.linalg.norm(u)).astype(np.float32)
        v = r.standard_normal(128); v = (v / np.linalg.norm(v)).astype(np.float32)
        rm = (0.3 * r.standard_normal(C)).astype(np.float32)
        rv = r.uniform(0.3, 2.0, C).astype(np.float32)
        z = F.linear(spk_noise, torch.from_numpy(fw).double(), torch.from_numpy(fb).double())
        w_sn, _u2, _v2 = O.spectral_norm_weight(torch.from_numpy(w).double(), torch.from_numpy(u).double(), torch.from_numpy(v).double(), False)
        gbw = F.linear(z, w_sn, torch.from_numpy(b).double())
        rstd = 1.0 / torch.sqrt(torch.from_numpy(rv).double() + 1e-5)
        aw = gbw[:, :C] * rstd
        want_a.append(aw); want_s.append(gbw[:, C:] - aw * torch.from_numpy(rm).double())
        fc_w.append(_t(fw, dev)); fc_b.append(_t(fb, dev)); sn_w.append(_t(w, dev)); sn_b.append(_t(b, dev))
        sn_u.append(_t(u, dev)); sn_v.append(_t(v, dev)); rms.append(_t(rm, dev)); rvs.append(_t(rv, dev))
    n = len(Cs)
    u0 = [t.clone() for t in sn_u]
    sg = torch.full((n,), float('nan'), device=dev)
    hipops.cond_sigma(sn_w, sn_u, sn_v, sg, training=False)
    a_out = [torch.full((B, C), float('nan'), device=dev) for C in Cs]
    s_out = [torch.full((B, C), float('nan'), device=dev) for C in Cs]
    hipops.cond_affine_eval(_t(spk, dev), _t(nz, dev), fc_w, fc_b, sn_w, sn_b, sg, rms, rvs, [1e-5] * n, a_out, s_out)
    assert all(torch.equal(a, b) for a, b in zip(sn_u, u0))
    # the path it replaces
    gb = [torch.empty((B, 2 * C), device=dev) for C in Cs]
    z_ws = torch.empty((n * B * 128,), device=dev)
    sg2 = torch.empty((n,), device=dev)
    hipops.cond_gamma_beta(_t(spk, dev), _t(nz, dev), fc_w, fc_b, sn_w, sn_b, sn_u, sn_v, gb, z_ws, sg2, False)
    assert torch.equal(sg, sg2)
    for i, C in enumerate(Cs):
        a2, s2 = torch.empty((B, C), device=dev), torch.empty((B, C), device=dev)
        hipops.bn_finalize(None, gb[i], rms[i], rvs[i], None, a2, s2, training=False, eps=1e-5)
        sc = max(1.0, want_a[i].abs().max().item(), want_s[i].abs().max().item())
        assert (a_out[i].cpu().double() - want_a[i]).abs().max().item() <= 2e-5 * sc
        assert (s_out[i].cpu().double() - want_s[i]).abs().max().item() <= 2e-5 * sc
        assert (a_out[i] - a2).abs().max().item() <= 2e-6 * sc and (s_out[i] - s2).abs().max().item() <= 2e-6 * sc


@pytest.mark.parametrize('B,C,L', [(3, 256, 250), (2, 16, 100000), (1, 32, 1), (4, 64, 4097)])
def test_bn_stats_and_finalize(dev, B, C, L):
    from wavthruvec_pytorch_amd import hipops
    r = _rng(7)
    x = (r.standard_normal((B, C, L)) * 2 + 0.5).astype(np.float32)
    gbv = r.standard_normal((B, 2 * C), dtype=np.float32)
    rm0 = (0.1 * r.standard_normal(C)).astype(np.float32)
    rv0 = r.uniform(0.5, 1.5, C).astype(np.float32)
    tx = torch.from_numpy(x)
    stats = torch.empty((2 * C + 1,), device=dev, dtype=torch.float64)
    part = torch.empty((2 * C * 64,), device=dev, dtype=torch.float64)
    xd = _t(x, dev)
    hipops.bn_stats(xd, stats, part)
    st = stats.cpu()
    s1 = tx.double().sum(dim=(0, 2)); s2 = tx.double().pow(2).sum(dim=(0, 2))
    assert (st[:C] - s1).abs().max().item() <= 1e-6 * max(1.0, s1.abs().max().item()) * 10
    assert (st[C:2 * C] - s2).abs().max().item() <= 1e-6 * s2.abs().max().item()
    assert st[2 * C].item() == B * L
    for training in (True, False):
        rm, rv = _t(rm0, dev), _t(rv0, dev)
        nbt = torch.tensor(7, device=dev, dtype=torch.long)
        a = torch.empty((B, C), device=dev); s = torch.empty((B, C), device=dev)
        hipops.bn_finalize(stats if training else None, _t(gbv, dev), rm, rv, nbt, a, s, training=training)
        xhat, rm_w, rv_w = O.batch_norm_no_affine(tx, torch.from_numpy(rm0), torch.from_numpy(rv0), training)
        gamma, beta = torch.from_numpy(gbv).chunk(2, 1)
        want = gamma[:, :, None] * xhat + beta[:, :, None]
        got = hipops.affine_apply(xd, a, s, torch.empty_like(xd)).cpu()
        # var == 0 (one value per channel) makes rstd = 1/sqrt(eps) = 316: the folded a*x + s form then carries
        # |a*x| * 2^-24 of rounding where the reference's (x - mean) is exactly 0
        tol = 2e-5 if B * L > 1 else 2e-4
        assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
        assert (rm.cpu() - rm_w).abs().max().item() <= 1e-6
        assert (rv.cpu() - rv_w).abs().max().item() <= 1e-5
        assert nbt.item() == (8 if training else 7)


@pytest.mark.parametrize('B,C,L,k', [(2, 16, 8000, 7), (1, 16, 3, 7), (2, 16, 1025, 7), (2, 8, 500, 3), (3, 16, 4, 7),
                                     (2, 16, 1000, 9), (1, 16, 640, 11)])
def test_conv_post_tanh(dev, B, C, L, k):
    from wavthruvec_pytorch_amd import hipops
    r = _rng(8)
    x = r.standard_normal((B, C, L), dtype=np.float32)
    w = (r.standard_normal((1, C, k)) / np.sqrt(C * k)).astype(np.float32)
    bias = r.standard_normal(1).astype(np.float32)
    want = torch.tanh(F.conv1d(F.leaky_relu(torch.from_numpy(x)), torch.from_numpy(w), torch.from_numpy(bias), padding=(k - 1) // 2))
    out = torch.empty((B, 1, L), device=dev)
    hipops.conv_post_tanh(_t(x, dev), _t(_relayout(torch.from_numpy(w)).numpy(), dev), _t(bias, dev), out, k=k, slope=0.01)
    assert (out.cpu() - want).abs().max().item() <= 2e-6


@pytest.mark.parametrize('B,C,L,k', [(2, 16, 8000, 7), (3, 16, 8, 7), (2, 16, 1024, 7), (1, 16, 12296, 9), (2, 8, 5000, 3), (1, 16, 1032, 1),
                                     (2, 16, 4100, 5), (3, 16, 4, 7), (2, 16, 1025, 7), (1, 16, 640, 11), (2, 32, 512, 7)])
def test_conv_post_tanh_bf16_input(dev, B, C, L, k):
    """The tail on bf16 activations (v2w_conv_post_tanh_bf16in).  C = 16 / 8, k <= 9, L % 8 == 0: the Toeplitz-MFMA kernel
    (v2w_conv_post_bf16.hip: weights as hi + lo bf16, relu on packed bf16, v_exp / v_rcp tanh) - rows shorter than a tile, rows ending
    inside a wave's tiles, several jobs per row; the other shapes: the vector-ALU kernels.  Reference: fp64 on the same bf16 values."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(8)
    x = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32) * 2).bfloat16()
    w = (r.standard_normal((1, C, k)) / np.sqrt(C * k)).astype(np.float32)
    bias = r.standard_normal(1).astype(np.float32) * 0.3
    want = torch.tanh(F.conv1d(F.leaky_relu(x.double(), 0.01), torch.from_numpy(w).double(), torch.from_numpy(bias).double(), padding=(k - 1) // 2))
    out = torch.full((B, 1, L), float('nan'), device=dev)
    hipops.conv_post_tanh(x.to(dev), _t(_relayout(torch.from_numpy(w)).numpy(), dev), _t(bias, dev), out, k=k, slope=0.01)
    err = (out.cpu().double() - want).abs().max().item()
    # the dominant (1 - slope) relu term carries 16 weight mantissa bits, the slope-scaled 1 % term 8: 2^-9 of 0.01 * sum |w x| (~3) here;
    # the rounding of the bf16 activations themselves is 50 times that
    assert err <= 2e-4, err


def test_conditional_batchnorm_module(dev):
    """`ConditionalBatchNorm1d(num_features).forward(inputs, noise)` standalone (reference modules.py:32-40 smoke shape)."""
    from wavthruvec_pytorch_amd import ConditionalBatchNorm1d
    torch.manual_seed(0)
    m = ConditionalBatchNorm1d(64)
    x = torch.randn(4, 64, 80); z = torch.randn(4, 128)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    xhat, rm, rv = O.batch_norm_no_affine(x, sd['batch_nrom.running_mean'], sd['batch_nrom.running_var'], True)
    w_sn, u2, v2 = O.spectral_norm_weight(sd['layer.weight_orig'], sd['layer.weight_u'], sd['layer.weight_v'], True)
    gamma, beta = F.linear(z, w_sn, sd['layer.bias']).chunk(2, 1)
    want = gamma[:, :, None] * xhat + beta[:, :, None]
    m = m.to(dev).train()
    got = m(x.to(dev), z.to(dev)).cpu()
    assert got.shape == (4, 64, 80)
    assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    assert (m.layer.weight_u.cpu() - u2).abs().max().item() <= 1e-6
    assert (m.batch_nrom.running_mean.cpu() - rm).abs().max().item() <= 1e-6
    assert m.batch_nrom.num_batches_tracked.item() == 1


def test_fold_pack_batch_equals_per_layer_path(dev):
    """v2w_fold_pack_batch (two launches for all layers) == v2w_wn_fold_* followed by v2w_pack_mfma, layer by layer."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(9)
    specs = [(768, 512, 7, 1, False), (256, 256, 11, 1, False), (16, 16, 3, 1, False), (32, 32, 7, 1, False),
             (512, 256, 11, 5, True), (32, 16, 4, 2, True), (512, 256, 16, 8, True), (64, 64, 3, 1, False)]
    layers, want = [], []
    for ci, co, k, u, tr in specs:
        shape = (ci, co, k) if tr else (co, ci, k)
        v = _t(r.standard_normal(shape, dtype=np.float32), dev)
        g = None if (ci, co) == (64, 64) else _t((1 + 0.1 * r.standard_normal((shape[0], 1, 1))).astype(np.float32), dev)
        wp = torch.full((k * ci * co,), float('nan'), device=dev)
        layers.append((v, g, wp, ci, co, k, u, tr))
        wf = (hipops.fold_convt_weight if tr else hipops.fold_conv_weight)(v, g)
        want.append(hipops.pack_mfma(wf, u=u))
    plan = hipops.FoldPlan(layers, dev)
    plan.run()
    for (v, g, wp, *_), w in zip(layers, want):
        assert (wp - w).abs().max().item() <= 2e-7 * w.abs().max().item()


def test_conv1d_multi_equals_sequential(dev):
    """Three branches (k = 3, 7, 11) in one launch, and the explicit-addend form of the running sum."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(10)
    B, C, L = 2, 64, 700
    x = _t(r.standard_normal((B, C, L), dtype=np.float32), dev)
    ia = _t((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32), dev)
    is_ = _t((0.3 * r.standard_normal((B, C))).astype(np.float32), dev)
    ws, bs = {}, {}
    for k in (3, 7, 11):
        w = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
        ws[k] = (_t(_relayout(w).numpy(), dev), w)
        bs[k] = _t(r.standard_normal(C).astype(np.float32), dev)
    seq, multi = {}, {}
    probs = []
    for k in (11, 7, 3):
        wf = ws[k][0]
        wp = hipops.pack_mfma(wf)
        seq[k] = torch.empty((B, C, L), device=dev)
        multi[k] = torch.full((B, C, L), float('nan'), device=dev)
        kw = dict(k=k, dil=3 if k > 3 else 1, slope=0.1, in_affine=(ia, is_), res=x, res_affine=(ia, is_), wp=wp)
        hipops.conv1d(x, wf, bs[k], seq[k], **kw)
        probs.append((x, wf, bs[k], multi[k], kw))
    hipops.conv1d_multi(probs)
    for k in (3, 7, 11):
        assert torch.equal(seq[k], multi[k]), k
    # ((o0 + o1) + value) / 3 with explicit addends == the accumulate chain
    wf = ws[11][0]; wp = hipops.pack_mfma(wf)
    chain = seq[3].clone()
    chain += seq[7]
    hipops.conv1d(x, wf, bs[11], chain, k=11, dil=1, slope=0.1, res=x, accumulate=True, out_div=3.0, wp=wp)
    fused = torch.empty_like(chain)
    hipops.conv1d(x, wf, bs[11], fused, k=11, dil=1, slope=0.1, res=x, add=[seq[3], seq[7]], out_div=3.0, wp=wp)
    assert torch.equal(chain, fused)
    direct = torch.empty_like(chain)
    hipops.conv1d(x, wf, bs[11], direct, k=11, dil=1, slope=0.1, res=x, add=[seq[3], seq[7]], out_div=3.0,
                  algo=hipops.ALGO_DIRECT)
    assert (direct - fused).abs().max().item() <= 2e-5


@pytest.mark.parametrize('C,L,k,d1,d2,mode', [(32, 1000, 11, 1, 3, 0), (32, 900, 3, 1, 3, 0), (16, 3000, 7, 1, 3, 0),
                                               (16, 250, 11, 5, 1, 1), (32, 40, 7, 3, 1, 1), (16, 4, 3, 1, 3, 0)])
def test_resblock_pair_fused_equals_two_convs(dev, C, L, k, d1, d2, mode):
    """The fused pair kernel (intermediate in LDS) vs the same two convs as separate launches, and vs torch."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(11)
    B = 2
    x = _t(r.standard_normal((B, C, L), dtype=np.float32), dev)
    ia = _t((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32), dev)
    is_ = _t((0.3 * r.standard_normal((B, C))).astype(np.float32), dev)
    w1 = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
    w2 = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
    b1 = _t(r.standard_normal(C).astype(np.float32), dev); b2 = _t(r.standard_normal(C).astype(np.float32), dev)
    a0 = _t(r.standard_normal((B, C, L), dtype=np.float32), dev); a1 = _t(r.standard_normal((B, C, L), dtype=np.float32), dev)
    wf1, wf2 = _t(_relayout(w1).numpy(), dev), _t(_relayout(w2).numpy(), dev)
    wp1, wp2 = hipops.pack_mfma(wf1), hipops.pack_mfma(wf2)
    aff = (ia, is_)
    # separate launches
    t1 = torch.empty((B, C, L), device=dev); ref = torch.empty((B, C, L), device=dev)
    if mode == 0:
        hipops.conv1d(x, None, b1, t1, k=k, dil=d1, slope=0.1, in_affine=aff, res=x, res_affine=aff, wp=wp1)
        hipops.conv1d(t1, None, b2, ref, k=k, dil=d2, slope=0.1, res=t1, add=[a0, a1], out_div=3.0, wp=wp2)
    else:
        hipops.conv1d(x, None, b1, t1, k=k, dil=d1, slope=0.1, in_affine=aff, wp=wp1)
        hipops.conv1d(t1, None, b2, ref, k=k, dil=d2, slope=0.1, res=x, res_affine=aff, add=[a0, a1], out_div=3.0, wp=wp2)
    out = torch.full((B, C, L), float('nan'), device=dev)
    ok = hipops.resblock_pair_multi([dict(x=x, in_affine=aff, wp1=wp1, b1=b1, wp2=wp2, b2=b2, out=out, k=k, dil1=d1, dil2=d2,
                                          res_mode=mode, slope=0.1, add=[a0, a1], out_div=3.0)])
    assert ok
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() <= 1e-6
    # against stock torch on the CPU
    xin = (ia[:, :, None] * x + is_[:, :, None]).cpu()
    c1 = F.conv1d(F.leaky_relu(xin, 0.1), w1, b1.cpu(), padding=d1 * (k - 1) // 2, dilation=d1)
    tt = c1 + xin if mode == 0 else c1
    c2 = F.conv1d(F.leaky_relu(tt, 0.1), w2, b2.cpu(), padding=d2 * (k - 1) // 2, dilation=d2)
    want = ((a0.cpu() + a1.cpu()) + (c2 + (tt if mode == 0 else xin))) / 3.0
    assert (out.cpu() - want).abs().max().item() <= 3e-5


@pytest.mark.parametrize('C,L', [(32, 1000), (16, 3000), (16, 4), (32, 252)])
def test_resblock2_stage_fused_equals_branchwise(dev, C, L):
    """One kernel for the whole ResBlock2 residual section of a stage vs the per-branch launches: same branch-sum order; the fused
    kernel starts its accumulators AT the bias (the per-layer kernel adds it last), so the two differ by a few ulps of O(5) values."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(12)
    B = 2
    x = _t(r.standard_normal((B, C, L), dtype=np.float32), dev)
    aff = (_t((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32), dev), _t((0.3 * r.standard_normal((B, C))).astype(np.float32), dev))
    branches, outs = [], []
    for k in (3, 7, 11):
        w1 = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
        w2 = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
        branches.append(dict(wp1=hipops.pack_mfma(_t(_relayout(w1).numpy(), dev)), b1=_t(r.standard_normal(C).astype(np.float32), dev),
                             wp2=hipops.pack_mfma(_t(_relayout(w2).numpy(), dev)), b2=_t(r.standard_normal(C).astype(np.float32), dev),
                             k=k, dil1=1, dil2=3, w1=w1, w2=w2))
    # per-branch path: two convs each, the last one adds the other two and divides
    for j, br in enumerate(branches):
        t1 = torch.empty((B, C, L), device=dev); o = torch.empty((B, C, L), device=dev)
        hipops.conv1d(x, None, br['b1'], t1, k=br['k'], dil=1, slope=0.1, in_affine=aff, res=x, res_affine=aff, wp=br['wp1'])
        extra = dict(add=outs[:2], out_div=3.0) if j == 2 else {}
        hipops.conv1d(t1, None, br['b2'], o, k=br['k'], dil=3, slope=0.1, res=t1, wp=br['wp2'], **extra)
        outs.append(o)
    got = torch.full((B, C, L), float('nan'), device=dev)
    assert hipops.resblock2_stage(x, aff, branches, got, slope=0.1, out_div=3.0)
    assert torch.isfinite(got).all()
    assert (got - outs[2]).abs().max().item() <= 4e-6
    xin = (aff[0][:, :, None] * x + aff[1][:, :, None]).cpu()
    want = None
    for br in branches:
        t1 = xin + F.conv1d(F.leaky_relu(xin, 0.1), br['w1'], br['b1'].cpu(), padding=(br['k'] - 1) // 2)
        rj = t1 + F.conv1d(F.leaky_relu(t1, 0.1), br['w2'], br['b2'].cpu(), padding=3 * (br['k'] - 1) // 2, dilation=3)
        want = rj if want is None else want + rj
    assert (got.cpu() - want / 3.0).abs().max().item() <= 3e-5


@pytest.mark.parametrize('L,kp', [(3000, 7), (218, 7), (5, 7), (1001, 7), (437, 3), (2048, 9), (64, 1)])
def test_resblock2_stage_f32_with_the_fused_tail(dev, L, kp):
    """v2w_resblock2_stage_fwd with post_out (ABI v29): the residual section of the last (16-channel) stage AND the generator's tail
    leaky_relu(0.01) -> conv_post (16 -> 1, kp taps) -> tanh (models.py:143-145) in one exact-fp32 kernel; the stage's output is not written.
    Against the two-kernel path (the same stage kernel without the tail + v2w_conv_post_tanh) and stock torch; lengths that are not
    multiples of 4 (scalar stores), shorter than a window, several windows with the 3-position tap halo across their seams."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(40 + L + kp)
    B, C = 2, 16
    x = _t(r.standard_normal((B, C, L), dtype=np.float32), dev)
    aff = (_t((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32), dev), _t((0.3 * r.standard_normal((B, C))).astype(np.float32), dev))
    branches = []
    for k in (3, 7, 11):
        w1 = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
        w2 = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
        branches.append(dict(wp1=hipops.pack_mfma(_t(_relayout(w1).numpy(), dev)), b1=_t(r.standard_normal(C).astype(np.float32), dev),
                             wp2=hipops.pack_mfma(_t(_relayout(w2).numpy(), dev)), b2=_t(r.standard_normal(C).astype(np.float32), dev),
                             k=k, dil1=1, dil2=3, w1=w1, w2=w2))
    wpost = torch.from_numpy((r.standard_normal((1, C, kp)) / np.sqrt(C * kp)).astype(np.float32))
    bpost = _t(r.standard_normal(1).astype(np.float32), dev)
    wf_post = _t(_relayout(wpost).numpy(), dev)               # [kp][C][1]
    stage = torch.full((B, C, L), float('nan'), device=dev)
    assert hipops.resblock2_stage(x, aff, branches, stage, slope=0.1, out_div=3.0)
    y2 = torch.full((B, 1, L), float('nan'), device=dev)
    hipops.conv_post_tanh(stage, wf_post, bpost, y2, k=kp, slope=0.01)
    y = torch.full((B, 1, L), float('nan'), device=dev)
    assert hipops.resblock2_stage(x, aff, branches, None, slope=0.1, out_div=3.0, post=(wf_post, bpost, y, kp, 0.01))
    assert torch.isfinite(y).all(), 'positions left unwritten'
    assert (y - y2).abs().max().item() <= 2e-6
    xin = (aff[0][:, :, None] * x + aff[1][:, :, None]).cpu()
    tot = None
    for br in branches:
        t1 = xin + F.conv1d(F.leaky_relu(xin, 0.1), br['w1'], br['b1'].cpu(), padding=(br['k'] - 1) // 2)
        rj = t1 + F.conv1d(F.leaky_relu(t1, 0.1), br['w2'], br['b2'].cpu(), padding=3 * (br['k'] - 1) // 2, dilation=3)
        tot = rj if tot is None else tot + rj
    want = torch.tanh(F.conv1d(F.leaky_relu(tot / 3.0, 0.01), wpost, bpost.cpu(), padding=(kp - 1) // 2))
    assert (y.cpu() - want).abs().max().item() <= 3e-5
    # the 32-channel stage has no tail form: declined, nothing launched
    x32 = torch.zeros((1, 32, 64), device=dev)
    br32 = [dict(wp1=branches[0]['wp1'], b1=None, wp2=branches[0]['wp2'], b2=None, k=3, dil1=1, dil2=3)]
    assert hipops.resblock2_stage(x32, None, br32, None, slope=0.1, out_div=1.0, post=(wf_post, bpost, torch.empty((1, 1, 64), device=dev), kp, 0.01)) is False


@pytest.mark.parametrize('C,L,ks,d1,d2', [(32, 1000, (3, 7, 11), 1, 3), (16, 3000, (3, 7, 11), 1, 3), (32, 200, (3, 7, 11), 1, 3), (16, 37, (3, 5), 2, 1),
                                         (32, 517, (11, 7, 3), 3, 1), (16, 2048, (3, 7, 11), 1, 3)])
def test_resblock2_stage_input_gradient_in_one_kernel(dev, C, L, ks, d1, d2):
    """v2w_resblock2_stage_fwd in its input-gradient form (ABI v31: v2w_stage_args::bwd_*): the backward of a narrow stage's residual section
    (models.py:135-141) - dt1_j of every branch and the stage's dx - from ONE kernel, against torch autograd through the section.  Lengths
    that are not multiples of 4, shorter than a window, several windows (the dt1_j of a position is written by the tile that owns it)."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(50 + L + C)
    B, nk = 2, len(ks)
    xr = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32))
    a = torch.from_numpy((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32))
    s = torch.from_numpy((0.3 * r.standard_normal((B, C))).astype(np.float32))
    dout = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32))
    x = (a[:, :, None] * xr + s[:, :, None]).requires_grad_(True)
    ws, t1s, tot = [], [], None
    for k in ks:
        w1 = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
        w2 = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
        t1 = x + F.conv1d(F.leaky_relu(x, 0.1), w1, None, padding=d1 * (k - 1) // 2, dilation=d1)
        t1.retain_grad()
        rj = t1 + F.conv1d(F.leaky_relu(t1, 0.1), w2, None, padding=d2 * (k - 1) // 2, dilation=d2)
        tot = rj if tot is None else tot + rj
        ws.append((w1, w2)); t1s.append(t1)
    (tot / nk).backward(dout)
    branches = [dict(wp1=hipops.pack_mfma_dgrad(_t(_relayout(w2).numpy(), dev)), b1=None, wp2=hipops.pack_mfma_dgrad(_t(_relayout(w1).numpy(), dev)), b2=None,
                     k=k, dil1=d2, dil2=d1) for k, (w1, w2) in zip(ks, ws)]
    inv, zero = torch.full((B, C), 1.0 / nk, device=dev), torch.zeros((B, C), device=dev)
    dt1 = [torch.full((B, C, L), float('nan'), device=dev) for _ in ks]
    dx = torch.full((B, C, L), float('nan'), device=dev)
    rows = hipops.resblock2_stage_bwd_rows(B, C, L, list(ks), [d2] * nk, [d1] * nk)
    assert rows > 0
    rsum = [torch.full((rows * C * 2,), float('nan'), device=dev) for _ in ks]
    assert hipops.resblock2_stage(dout.to(dev), (inv, zero), branches, dx, slope=1.0, out_div=0.0,
                                  bwd=([t.detach().to(dev) for t in t1s], dt1, xr.to(dev), (a.to(dev), s.to(dev)), 0.1, rsum))
    for j in range(nk):
        assert torch.isfinite(dt1[j]).all(), 'positions left unwritten'
        assert (dt1[j].cpu() - t1s[j].grad).abs().max().item() <= 3e-5 * max(1.0, t1s[j].grad.abs().max().item()), j
        # the (tile, wave) partial sums add up to the bias gradient of conv1_j
        st = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        hipops.bn_reduce_partials(rsum[j], rows, C, B * L, st)
        want_b = t1s[j].grad.double().sum(dim=(0, 2))
        assert (st[:C].cpu() - want_b).abs().max().item() <= 1e-4 * max(1.0, want_b.abs().max().item()), j
    assert torch.isfinite(dx).all()
    assert (dx.cpu() - x.grad).abs().max().item() <= 5e-5 * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize('B,C,L,k', [(2, 16, 4100, 7), (3, 16, 1024, 7), (1, 16, 8, 7), (2, 16, 1030, 7), (2, 16, 2052, 5), (2, 32, 1000, 7), (2, 8, 640, 7)])
def test_tail_backward_matches_autograd(dev, B, C, L, k):
    """v2w_tail_bwd: the backward of leaky_relu(0.01) -> conv_post -> tanh (models.py:143-145) - dx, the weight gradient and dp (whose sum is the
    bias gradient) against torch autograd.  C = 16, k = 7 at L % 4 == 0 runs the one-pass kernel (ABI v32: x read once for both gradients,
    several 1 024-position tiles with their seams, a length below one tile); everything else the per-tap kernels."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(60 + L + C + k)
    x = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32)).requires_grad_(True)
    w = torch.from_numpy((r.standard_normal((1, C, k)) / np.sqrt(C * k)).astype(np.float32)).requires_grad_(True)
    bias = torch.zeros(1, requires_grad=True)
    y = torch.tanh(F.conv1d(F.leaky_relu(x, 0.01), w, bias, padding=(k - 1) // 2))
    dy = torch.from_numpy(r.standard_normal((B, 1, L), dtype=np.float32))
    y.backward(dy)
    dx, dwf, dp = hipops.tail_backward(_t(dy.numpy(), dev), _t(y.detach().numpy(), dev), _t(x.detach().numpy(), dev), _t(_relayout(w.detach()).numpy(), dev),
                                       k=k, slope=0.01)
    assert torch.isfinite(dx).all() and (dx.cpu() - x.grad).abs().max().item() <= 2e-5 * max(1.0, x.grad.abs().max().item())
    want_w = w.grad.permute(2, 1, 0)                                   # [k][C][1]
    assert (dwf.cpu() - want_w).abs().max().item() <= 1e-4 * max(1.0, want_w.abs().max().item())
    assert abs(dp.sum().item() - bias.grad.item()) <= 1e-3 * max(1.0, abs(bias.grad.item()))


@pytest.mark.parametrize('B,C,L,k,dil', [(2, 256, 300, 11, 3), (2, 64, 700, 7, 1), (3, 32, 1000, 3, 3), (2, 16, 3000, 11, 1)])
def test_conv1d_dgrad_building_block(dev, B, C, L, k, dil):
    """Backward through one ResBlock2 step  y = x + conv_{k,d}(lrelu(x)),  x = a*in + s  (first piece):
    dx = dy + lrelu'(x) * conv_{k,d}(dy; W^T flipped) is the SAME fused conv kernel run on dy with transposed-flipped weights,
    the leaky_relu derivative applied as an epilogue mask and dy as the residual.  Checked against torch autograd."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(13)
    xin = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32))
    a = torch.from_numpy((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32))
    s = torch.from_numpy((0.3 * r.standard_normal((B, C))).astype(np.float32))
    w = torch.from_numpy((r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
    dy = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32))
    x = (a[:, :, None] * xin + s[:, :, None]).requires_grad_(True)
    y = x + F.conv1d(F.leaky_relu(x, 0.1), w, None, padding=dil * (k - 1) // 2, dilation=dil)
    y.backward(dy)
    want = x.grad
    wf = _t(_relayout(w).numpy(), dev)
    wpT = hipops.pack_mfma(hipops.transpose_flip(wf))
    dyd = _t(dy.numpy(), dev)
    got = torch.full((B, C, L), float('nan'), device=dev)
    hipops.conv1d(dyd, None, None, got, k=k, dil=dil, slope=1.0, res=dyd, wp=wpT,
                  mask=(_t(xin.numpy(), dev), (_t(a.numpy(), dev), _t(s.numpy(), dev))), mask_slope=0.1)
    assert (got.cpu() - want).abs().max().item() <= 3e-5
    # direct kernel agrees
    got2 = torch.empty_like(got)
    hipops.conv1d(dyd, hipops.transpose_flip(wf), None, got2, k=k, dil=dil, slope=1.0, res=dyd, algo=hipops.ALGO_DIRECT,
                  mask=(_t(xin.numpy(), dev), (_t(a.numpy(), dev), _t(s.numpy(), dev))), mask_slope=0.1)
    assert (got2.cpu() - want).abs().max().item() <= 3e-5


@pytest.mark.parametrize('B,cin,cout,L,k,dil,u', [(2, 256, 256, 300, 11, 3, 1), (2, 64, 64, 1000, 7, 1, 1), (3, 32, 32, 777, 3, 5, 1),
                                                   (2, 16, 16, 3000, 11, 1, 1), (2, 768, 512, 50, 7, 1, 1),
                                                   (2, 512, 256, 50, 11, 1, 5), (2, 128, 64, 333, 8, 1, 4), (2, 64, 32, 500, 4, 1, 2),
                                                   (2, 32, 16, 1000, 4, 1, 2), (1, 512, 256, 17, 16, 1, 8),
                                                   # the 8-channel stage of a six-stage (x640) generator: 16-row tiles with the missing rows staged as 0
                                                   (2, 8, 8, 1999, 11, 3, 1), (2, 8, 8, 640, 3, 1, 1), (2, 16, 8, 700, 4, 1, 2),
                                                   # >= 512 staged items of a one-tile layer: 2 048 slabs, summed in two levels (wgrad_reduce_part_kernel)
                                                   (4, 32, 32, 16384, 3, 1, 1), (3, 16, 16, 22000, 7, 3, 1), (4, 32, 16, 16384, 4, 1, 2)])
def test_wgrad_matches_autograd(dev, B, cin, cout, L, k, dil, u):
    """dW of the fused [affine] -> lrelu -> Conv1d / ConvTranspose1d against torch autograd in fp64 (weights in [k][C_in][C_out] layout)."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(14)
    x = torch.from_numpy(r.standard_normal((B, cin, L), dtype=np.float32))
    a = torch.from_numpy((1 + 0.2 * r.standard_normal((B, cin))).astype(np.float32))
    s = torch.from_numpy((0.3 * r.standard_normal((B, cin))).astype(np.float32))
    act = F.leaky_relu(a[:, :, None].double() * x.double() + s[:, :, None].double(), 0.1)
    if u == 1:
        w = torch.from_numpy(r.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).requires_grad_(True)
        y = F.conv1d(act, w, None, padding=dil * (k - 1) // 2, dilation=dil)
    else:
        w = torch.from_numpy(r.standard_normal((cin, cout, k)) / np.sqrt(cin * k)).requires_grad_(True)
        y = F.conv_transpose1d(act, w, None, stride=u, padding=(k - u) // 2)
    dy = torch.from_numpy(r.standard_normal(tuple(y.shape), dtype=np.float32))
    y.backward(dy.double())
    want = (w.grad.permute(2, 1, 0) if u == 1 else w.grad.permute(2, 0, 1)).float()      # -> [k][C_in][C_out]
    got = hipops.wgrad(_t(x.numpy(), dev), _t(dy.numpy(), dev), k=k, dil=dil, u=u, slope=0.1,
                       x_affine=(_t(a.numpy(), dev), _t(s.numpy(), dev))).cpu()
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5 * max(1.0, scale) * 10


@pytest.mark.gpu
@pytest.mark.parametrize('stored', ['f32', 'bf16'])
@pytest.mark.parametrize('B,C,L,k,dil', [(2, 16, 1024, 3, 1), (3, 16, 520, 7, 3), (2, 16, 2048, 11, 3), (2, 32, 1000, 3, 3), (3, 32, 648, 7, 1),
                                         (1, 16, 8, 3, 1), (1, 32, 16, 11, 3), (1, 64, 136, 7, 1), (5, 128, 8, 3, 1),      # below one staged item, one item + 8, B > splits

                                         (2, 32, 1024, 11, 3), (2, 64, 512, 3, 1), (2, 64, 328, 7, 3), (1, 64, 640, 11, 1), (2, 128, 264, 11, 3),
                                         (1, 256, 136, 7, 5), (2, 64, 512, 5, 1), (2, 32, 512, 9, 1), (2, 64, 256, 9, 2)])
def test_wgrad_bf16_matches_the_products_of_rounded_operands(dev, B, C, L, k, dil, stored):
    """v2w_wgrad_bf16: dW of [affine] -> lrelu -> Conv1d from bf16 operands with fp32 accumulation.  The reference value is torch autograd (fp64)
    on the operands rounded the way the kernel rounds them - act(x) computed in fp32 and rounded once, dy rounded - so only the fp32 summation
    order differs.  `stored`: the tensors handed over are fp32 (rounded while staged) or already bf16."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(31)
    x = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32))
    dy = torch.from_numpy(r.standard_normal((B, C, L), dtype=np.float32))
    a = torch.from_numpy((1 + 0.2 * r.standard_normal((B, C))).astype(np.float32))
    s = torch.from_numpy((0.3 * r.standard_normal((B, C))).astype(np.float32))
    if stored == 'bf16':
        x, dy = x.bfloat16(), dy.bfloat16()
    act = F.leaky_relu(torch.addcmul(s[:, :, None], a[:, :, None], x.float()), 0.1).bfloat16().double()     # fma, like the kernel
    w = torch.zeros((C, C, k), dtype=torch.float64, requires_grad=True)
    F.conv1d(act, w, None, padding=dil * (k - 1) // 2, dilation=dil).backward(dy.bfloat16().double())
    want = w.grad.permute(2, 1, 0)
    got = hipops.wgrad_bf16(x.to(dev), dy.to(dev), k=k, dil=dil, slope=0.1, x_affine=(a.to(dev), s.to(dev)))
    assert got is not None
    err = (got.cpu().double() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err
    # no affine: the plain activated signal
    w2 = torch.zeros((C, C, k), dtype=torch.float64, requires_grad=True)
    F.conv1d(F.leaky_relu(x.float(), 0.1).bfloat16().double(), w2, None, padding=dil * (k - 1) // 2, dilation=dil).backward(dy.bfloat16().double())
    got2 = hipops.wgrad_bf16(x.to(dev), dy.to(dev), k=k, dil=dil, slope=0.1)
    assert (got2.cpu().double() - w2.grad.permute(2, 1, 0)).abs().max().item() <= 2e-5 * max(1.0, w2.grad.abs().max().item())
    # deterministic
    assert torch.equal(got2, hipops.wgrad_bf16(x.to(dev), dy.to(dev), k=k, dil=dil, slope=0.1))


@pytest.mark.gpu
def test_wgrad_bf16_declines_shapes_it_has_no_kernel_for(dev):
    from wavthruvec_pytorch_amd import hipops
    x = torch.randn(2, 48, 256, device=dev)
    assert hipops.wgrad_bf16(x, x, k=3) is None                    # 48 channels
    x = torch.randn(2, 32, 252, device=dev)
    assert hipops.wgrad_bf16(x, x, k=3) is None                    # rows not 16-byte aligned in bf16
    x = torch.randn(2, 32, 256, device=dev)
    assert hipops.wgrad_bf16(x, x, k=11, dil=7) is None            # halo beyond the staged tile
    assert hipops.wgrad_bf16(x, x, k=4) is None


@pytest.mark.parametrize('B,cin,cout,L,k,u', [(2, 512, 256, 50, 11, 5), (2, 128, 64, 333, 8, 4), (2, 64, 32, 500, 4, 2),
                                             (2, 32, 16, 1000, 4, 2), (1, 512, 256, 17, 16, 8)])
def test_convt1d_dgrad_matches_autograd(dev, B, cin, cout, L, k, u):
    """dx of  y = ConvTranspose1d(lrelu(x)): one small Conv1d per output phase on the strided phase of dy, lrelu' as epilogue mask."""
    from wavthruvec_pytorch_amd import hipops
    r = _rng(15)
    x = torch.from_numpy(r.standard_normal((B, cin, L), dtype=np.float32)).requires_grad_(True)
    w = torch.from_numpy((r.standard_normal((cin, cout, k)) / np.sqrt(cin * k)).astype(np.float32))
    y = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, None, stride=u, padding=(k - u) // 2)
    dy = torch.from_numpy(r.standard_normal(tuple(y.shape), dtype=np.float32))
    y.backward(dy)
    wf = _t(w.permute(2, 0, 1).contiguous().numpy(), dev)
    out = torch.full((B, cin, L), float('nan'), device=dev)
    hipops.convt1d_dgrad(_t(dy.numpy(), dev), wf, out, k=k, u=u, mask=(_t(x.detach().numpy(), dev), None), mask_slope=0.1)
    assert (out.cpu() - x.grad).abs().max().item() <= 3e-5


@pytest.mark.parametrize('B,L', [(2, 8192), (3, 81920), (1, 5000)])
def test_mel_spectrogram_matches_oracle(dev, B, L):
    """mel_spectrogram (dataset.py:53-77) on the HIP path against the CPU restatement (torch.stft + the restated Slaney filterbank)."""
    from oracle import mel_oracle as M
    from wavthruvec_pytorch_amd.mel import mel_spectrogram, mel_filterbank
    y = torch.from_numpy(np.tanh(_rng(16).standard_normal((B, L))).astype(np.float32) * 0.9)
    want = M.mel_spectrogram(y, 1024, 80, 16000, 256, 1024, 0, None, dtype=torch.float64)
    got = mel_spectrogram(y.to(dev), 1024, 80, 16000, 256, 1024, 0, None).cpu()
    assert got.shape == want.shape == (B, 80, L // 256)
    assert (got.double() - want).abs().max().item() <= 2e-4        # log of fp32 sums of 513 magnitudes
    assert np.array_equal(mel_filterbank(16000, 1024, 80, 0, 8000), M.mel_filterbank(16000, 1024, 80, 0, 8000))
    with pytest.raises(NotImplementedError):
        mel_spectrogram(y.to(dev), 1024, 80, 16000, 256, 1024, 0, None, center=True)


@pytest.mark.parametrize('B,L', [(2, 8192), (1, 5000), (2, 20480)])
def test_mel_spectrogram_backward_matches_oracle_autograd(dev, B, L):
    """d mel_spectrogram / d y (the training loss back-propagates through it, train.py:172-174,204) against torch autograd through
    the fp64 CPU restatement: a smooth weighted sum, and the reference's own L1 loss against a target mel."""
    from oracle import mel_oracle as M
    from wavthruvec_pytorch_amd.mel import mel_spectrogram
    r = _rng(23)
    y0 = torch.from_numpy(np.tanh(r.standard_normal((B, L))).astype(np.float32) * 0.9)
    tgt = torch.from_numpy(np.tanh(r.standard_normal((B, L))).astype(np.float32) * 0.5)
    G = torch.from_numpy(r.standard_normal((B, 80, L // 256)).astype(np.float32))
    args = (1024, 80, 16000, 256, 1024, 0, None)
    yo = y0.double().requires_grad_(True)
    mo = M.mel_spectrogram(yo, *args, dtype=torch.float64)
    (mo * G.double()).sum().backward()
    yd = y0.to(dev).requires_grad_(True)
    md = mel_spectrogram(yd, *args)
    assert md.requires_grad and (md.detach().cpu().double() - mo.detach()).abs().max().item() <= 2e-4
    (md * G.to(dev)).sum().backward()
    scale = yo.grad.abs().max().item()
    assert (yd.grad.cpu().double() - yo.grad).abs().max().item() <= 2e-4 * scale
    # the reflect padding folds the edge gradients back: the first / last (n_fft - hop)/2 samples carry two contributions
    assert (yd.grad.cpu().double()[:, :400] - yo.grad[:, :400]).abs().max().item() <= 2e-4 * scale
    assert (yd.grad.cpu().double()[:, -400:] - yo.grad[:, -400:]).abs().max().item() <= 2e-4 * scale
    # train.py:204  loss_mel = F.l1_loss(y_mel, y_g_hat_mel) * 45  (sign() gradients: compared where the residual is not ~0)
    y_mel = M.mel_spectrogram(tgt, *args)
    yo.grad = None
    (torch.nn.functional.l1_loss(y_mel.double(), M.mel_spectrogram(yo, *args, dtype=torch.float64)) * 45).backward()
    yd.grad = None
    (torch.nn.functional.l1_loss(y_mel.to(dev), mel_spectrogram(yd, *args)) * 45).backward()
    assert (yd.grad.cpu().double() - yo.grad).abs().max().item() <= 1e-3 * yo.grad.abs().max().item()
    # no graph when the input does not ask for one
    assert not mel_spectrogram(y0.to(dev), *args).requires_grad


# ---------------------------------------------------------------------------------------------------------------------
# entry points of the discriminator path (include/vec2wav_hip.h: v2w_phase_split ... v2w_wgrad_groups), one by one
def _lib_stream(dev):
    from wavthruvec_pytorch_amd import _hip
    return _hip, _hip.load(), torch.cuda.current_stream(dev).cuda_stream


@pytest.mark.parametrize('B,C,Cg,L,inner,s', [(2, 8, 8, 37, 1, 2), (2, 12, 4, 50, 1, 4), (1, 6, 6, 20, 13, 3), (3, 4, 2, 7, 5, 3)])
def test_phase_split_and_merge(dev, B, C, Cg, L, inner, s):
    """v2w_phase_split against its definition (pitched rows, zero tails), v2w_phase_merge as its exact inverse on the valid part."""
    _hip, lib, st = _lib_stream(dev)
    r = _rng(3)
    U = -(-L // s)
    ip, op = (L * inner + 3) // 4 * 4, (U * inner + 3) // 4 * 4
    x = torch.zeros(B, C, ip)
    x[:, :, :L * inner] = torch.from_numpy(r.standard_normal((B, C, L * inner)).astype(np.float32))
    xd = x.to(dev)
    out = torch.full((B, s * C, op), float('nan'), device=dev)
    _hip.check(lib.v2w_phase_split(xd.data_ptr(), out.data_ptr(), B, C, Cg, L, inner, s, ip, op, st), 'v2w_phase_split')
    x4 = x[:, :, :L * inner].view(B, C, L, inner)
    want = torch.zeros(B, s * C, op)
    for c in range(C):
        for rr in range(s):
            rows = x4[:, c, rr::s, :]                                  # (B, n_u, inner)
            cs = ((c // Cg) * s + rr) * Cg + c % Cg
            want[:, cs, :rows.shape[1] * inner] = rows.reshape(B, -1)
    assert torch.equal(out.cpu(), want)
    back = torch.full((B, C, ip), float('nan'), device=dev)
    _hip.check(lib.v2w_phase_merge(out.data_ptr(), back.data_ptr(), B, C, Cg, L, inner, s, op, ip, st), 'v2w_phase_merge')
    assert torch.equal(back.cpu()[:, :, :L * inner], x[:, :, :L * inner])


@pytest.mark.parametrize('B,T,inner,s,k,pad', [(2, 1000, 13, 3, 5, 2), (1, 250, 19, 3, 5, 2), (3, 333, 1, 1, 15, 7)])
def test_unfold1_and_its_adjoint(dev, B, T, inner, s, k, pad):
    """v2w_unfold1 (right reflect pad to a multiple of `inner`, k shifted rows) against torch, v2w_fold1 as its adjoint:
    <unfold1(x), y> == <x, fold1(y)>."""
    _hip, lib, st = _lib_stream(dev)
    r = _rng(4)
    H = -(-T // inner)
    U = (H + 2 * pad - k) // s + 1
    P = (U * inner + 3) // 4 * 4
    x = torch.from_numpy(r.standard_normal((B, T)).astype(np.float32))
    out = torch.full((B, 16, P), float('nan'), device=dev)
    xd = x.to(dev)                        # (device copies are held in variables: a temporary would be freed before the launch)
    _hip.check(lib.v2w_unfold1(xd.data_ptr(), out.data_ptr(), B, T, H, inner, s, k, pad, 16, P, st), 'v2w_unfold1')
    xp = F.pad(x.unsqueeze(1), (0, H * inner - T), 'reflect').squeeze(1) if H * inner > T else x
    x2 = F.pad(xp.view(B, H, inner), (0, 0, pad, pad))                # zero rows above / below
    want = torch.zeros(B, 16, P)
    for j in range(k):
        rows = x2[:, j:j + s * (U - 1) + 1:s, :]                       # (B, U, inner)
        want[:, j, :U * inner] = rows.reshape(B, -1)
    assert torch.equal(out.cpu(), want)
    y = torch.zeros(B, 16, P)
    y[:, :k, :U * inner] = torch.from_numpy(r.standard_normal((B, k, U * inner)).astype(np.float32))
    dx = torch.full((B, T), float('nan'), device=dev)
    yd = y.to(dev)
    _hip.check(lib.v2w_fold1(yd.data_ptr(), dx.data_ptr(), B, T, H, inner, s, k, pad, 16, P, st), 'v2w_fold1')
    lhs = (want.double() * y.double()).sum().item()
    rhs = (x.double() * dx.cpu().double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))


@pytest.mark.parametrize('B,L', [(2, 100), (3, 101), (1, 7)])
def test_avgpool4_and_backward(dev, B, L):
    _hip, lib, st = _lib_stream(dev)
    x = torch.from_numpy(_rng(5).standard_normal((B, 1, L)).astype(np.float32)).requires_grad_(True)
    want = F.avg_pool1d(x, 4, 2, padding=2)
    g = torch.from_numpy(_rng(6).standard_normal(tuple(want.shape)).astype(np.float32))
    want.backward(g)
    out = torch.empty((B, L // 2 + 1), device=dev)
    xd, gd = x.detach().to(dev), g.to(dev)
    _hip.check(lib.v2w_avgpool4(xd.data_ptr(), out.data_ptr(), B, L, st), 'v2w_avgpool4')
    assert out.shape[1] == want.shape[2] and (out.cpu() - want.detach().squeeze(1)).abs().max().item() <= 1e-6
    dx = torch.empty((B, L), device=dev)
    _hip.check(lib.v2w_avgpool4_bwd(gd.data_ptr(), dx.data_ptr(), B, L, st), 'v2w_avgpool4_bwd')
    assert (dx.cpu() - x.grad.squeeze(1)).abs().max().item() <= 1e-6


@pytest.mark.parametrize('G,cig,cog,k,L', [(4, 32, 32, 5, 200), (16, 16, 16, 3, 96), (2, 64, 64, 7, 130), (16, 64, 64, 41, 64)])
def test_grouped_conv_slices_out_slope_and_wgrad_groups(dev, G, cig, cog, k, L):
    """One group per problem through in_ct / out_ct (four per launch), leaky_relu on the stored value (out_slope), the input
    gradient with the transposed tap-flipped weights and v2w_wgrad_groups - against torch's grouped Conv1d and its autograd."""
    from wavthruvec_pytorch_amd import hipops
    _hip, lib, st = _lib_stream(dev)
    r = _rng(7)
    B = 2
    x = torch.from_numpy(r.standard_normal((B, G * cig, L)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((r.standard_normal((G * cog, cig, k)) / np.sqrt(cig * k)).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(r.standard_normal(G * cog).astype(np.float32))
    z = F.conv1d(x, w, b, padding=(k - 1) // 2, groups=G)
    want = F.leaky_relu(z, 0.1)
    gz = torch.from_numpy(r.standard_normal(tuple(z.shape)).astype(np.float32))
    z.backward(gz)
    w4 = w.detach().view(G, cog, cig, k).permute(0, 3, 2, 1).contiguous().to(dev)          # [G][k][cig][cog]
    wp = hipops.pack_mfma_batch(w4)
    xd, bd = x.detach().to(dev), b.to(dev)
    out = torch.full((B, G * cog, L), float('nan'), device=dev)
    probs = [(xd, w4[g], bd[g * cog:(g + 1) * cog], out, dict(k=k, dil=1, slope=1.0, out_slope=0.1, wp=wp[g], group=(g, cig, cog)))
             for g in range(G)]
    for i in range(0, G, 4):
        hipops.conv1d_multi(probs[i:i + 4])
    assert (out.cpu() - want.detach()).abs().max().item() <= 3e-5
    # input gradient
    gzd = gz.to(dev)
    wT4 = w4.flip(1).transpose(2, 3).contiguous()
    wTp = hipops.pack_mfma_batch(wT4)
    dx = torch.full((B, G * cig, L), float('nan'), device=dev)
    for g in range(G):
        hipops.conv1d(gzd, wT4[g], None, dx, k=k, dil=1, slope=1.0, wp=wTp[g], group=(g, cog, cig))
    assert (dx.cpu() - x.grad).abs().max().item() <= 3e-5 * max(1.0, x.grad.abs().max().item())
    # weight gradient, all groups in one launch
    ns = lib.v2w_wgrad_group_slabs(B, cig, cog, L, G)
    assert 0 < ns <= lib.v2w_wgrad_slabs(B, cig, cog, L)
    slab = torch.empty((G * ns * k * cig * cog,), device=dev)
    dw = torch.full((G, k, cig, cog), float('nan'), device=dev)
    _hip.check(lib.v2w_wgrad_groups(xd.data_ptr(), gzd.data_ptr(), dw.data_ptr(), slab.data_ptr(), B, cig, cog, L, k, 1, -1, G, st),
               'v2w_wgrad_groups')
    ref = w.grad.view(G, cog, cig, k).permute(0, 3, 2, 1)
    assert (dw.cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('B,C,L,k,dil,tap0', [(2, 64, 100, 3, 1, 1), (3, 128, 57, 3, 13, 1), (1, 32, 40, 2, 5, 1)])
def test_cout1_conv_and_wgrad(dev, B, C, L, k, dil, tap0):
    """The C_out = 1 reduction kernel behind v2w_conv1d_fwd (conv_post of the discriminators) and v2w_cout1_wgrad."""
    from wavthruvec_pytorch_amd import hipops
    _hip, lib, st = _lib_stream(dev)
    r = _rng(8)
    x = torch.from_numpy(r.standard_normal((B, C, L)).astype(np.float32))
    w = torch.from_numpy((r.standard_normal((1, C, k)) / np.sqrt(C * k)).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(r.standard_normal(1).astype(np.float32))
    xp = F.pad(x, (tap0 * dil, dil * (k - 1) - tap0 * dil))
    want = F.conv1d(xp, w, b, dilation=dil)
    g = torch.from_numpy(r.standard_normal(tuple(want.shape)).astype(np.float32))
    want.backward(g)
    wf = w.detach().permute(2, 1, 0).contiguous().to(dev)
    out = torch.full((B, 1, L), float('nan'), device=dev)
    xd, bd, gd = x.to(dev), b.to(dev), g.to(dev)
    hipops.conv1d(xd, wf, bd, out, k=k, dil=dil, slope=1.0, pad_left=tap0 * dil)
    assert (out.cpu() - want.detach()).abs().max().item() <= 2e-5
    dwf = torch.full((k, C, 1), float('nan'), device=dev)
    _hip.check(lib.v2w_cout1_wgrad(xd.data_ptr(), gd.data_ptr(), dwf.data_ptr(), B, C, L, k, dil, tap0, st),
               'v2w_cout1_wgrad')
    assert (dwf.cpu() - w.grad.permute(2, 1, 0)).abs().max().item() <= 1e-4 * max(1.0, w.grad.abs().max().item())


def test_disc_dz_and_merge_and_zero_tail(dev):
    """v2w_disc_dz / v2w_disc_dz_merge (gradient sum x leaky_relu' of the ACTIVATED map, zero pitch tail) and v2w_zero_tail."""
    _hip, lib, st = _lib_stream(dev)
    r = _rng(9)
    B, C, L, inner, s, Cg = 2, 6, 10, 3, 2, 3
    valid, P = L * inner, (L * inner + 3) // 4 * 4
    f = torch.from_numpy(r.standard_normal((B, C, P)).astype(np.float32))
    g = torch.from_numpy(r.standard_normal((B, C, valid)).astype(np.float32))
    d = torch.from_numpy(r.standard_normal((B, C, P)).astype(np.float32))
    mask = torch.where(f[:, :, :valid] > 0, torch.tensor(1.0), torch.tensor(0.1))
    dz = torch.full((B, C, P), float('nan'), device=dev)
    fd, gd, dd = f.to(dev), g.to(dev), d.to(dev)
    rs = torch.full((B * C,), float('nan'), device=dev)
    _hip.check(lib.v2w_disc_dz(fd.data_ptr(), gd.data_ptr(), dd.data_ptr(), dz.data_ptr(), rs.data_ptr(), B * C, P, valid, 0.1, st),
               'v2w_disc_dz')
    db = torch.full((C,), float('nan'), device=dev)
    _hip.check(lib.v2w_rowsum_reduce(rs.data_ptr(), db.data_ptr(), B, C, st), 'v2w_rowsum_reduce')
    assert (db.cpu() - ((g + d[:, :, :valid]) * mask).sum(dim=(0, 2))).abs().max().item() <= 1e-5
    assert torch.equal(dz.cpu()[:, :, valid:], torch.zeros(B, C, P - valid))
    assert (dz.cpu()[:, :, :valid] - (g + d[:, :, :valid]) * mask).abs().max().item() <= 1e-6
    # merge form: d given as the phase-stacked gradient of a stride-s layer
    U = -(-L // s)
    dp = (U * inner + 3) // 4 * 4
    dxs = torch.from_numpy(r.standard_normal((B, s * C, dp)).astype(np.float32)).to(dev)
    merged = torch.full((B, C, P), float('nan'), device=dev)
    _hip.check(lib.v2w_phase_merge(dxs.data_ptr(), merged.data_ptr(), B, C, Cg, L, inner, s, dp, P, st), 'v2w_phase_merge')
    dz2 = torch.full((B, C, P), float('nan'), device=dev)
    _hip.check(lib.v2w_disc_dz_merge(fd.data_ptr(), gd.data_ptr(), dxs.data_ptr(), dz2.data_ptr(), None, B, C, Cg, L, inner, s, dp, P,
                                     0.1, st), 'v2w_disc_dz_merge')
    want = (g + merged.cpu()[:, :, :valid]) * mask
    assert (dz2.cpu()[:, :, :valid] - want).abs().max().item() <= 1e-6 and torch.equal(dz2.cpu()[:, :, valid:], torch.zeros(B, C, P - valid))
    t = torch.ones((B * C, P), device=dev)
    _hip.check(lib.v2w_zero_tail(t.data_ptr(), B * C, P, valid, st), 'v2w_zero_tail')
    assert torch.equal(t.cpu()[:, :valid], torch.ones(B * C, valid)) and torch.equal(t.cpu()[:, valid:], torch.zeros(B * C, P - valid))


@pytest.mark.gpu
@pytest.mark.parametrize('C,nt,B', [(256, 256, 32), (128, 480, 32), (32, 512, 300), (16, 1, 1), (24, 777, 5)])
def test_bn_reduce_finalize_is_the_two_calls_bit_for_bit(C, nt, B):
    """v2w_bn_reduce_finalize (ABI v34: one launch, one block per channel) == v2w_bn_reduce_partials + v2w_bn_finalize(training = 1): the sums array,
    (a, s), the running statistics and num_batches_tracked, bit for bit (B > 256: more samples than threads in the block)."""
    from wavthruvec_pytorch_amd import hipops
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(C * 11 + nt)
    part = torch.randn(nt, C, 2, generator=g).abs_().mul_(50).to(dev)
    part[..., 1] += part[..., 0] ** 2 / 100
    count = nt * 224
    gb = torch.randn(B, 2 * C, generator=g).to(dev)
    outs = []
    for fused in (False, True):
        rm, rv = torch.randn(C, generator=g).to(dev), torch.rand(C, generator=torch.Generator().manual_seed(3)).add_(0.5).to(dev)
        if outs:
            rm, rv = outs[0][5].clone(), outs[0][6].clone()
        rm0, rv0 = rm.clone(), rv.clone()
        nb = torch.full((), 41, dtype=torch.int64, device=dev)
        a, s = torch.full((B, C), float('nan'), device=dev), torch.full((B, C), float('nan'), device=dev)
        st = torch.full((2 * C + 1,), float('nan'), dtype=torch.float64, device=dev)
        if fused:
            hipops.bn_reduce_finalize(part, nt, count, st, gb, rm, rv, nb, a, s, momentum=0.1, eps=1e-5)
        else:
            hipops.bn_reduce_partials(part, nt, C, count, st)
            hipops.bn_finalize(st, gb, rm, rv, nb, a, s, training=True, momentum=0.1, eps=1e-5)
        outs.append((a, s, rm, rv, st, rm0, rv0, nb))
    for x, y in zip(outs[0][:5], outs[1][:5]):
        assert torch.isfinite(y.double()).all() and torch.equal(x, y)
    assert outs[0][7].item() == outs[1][7].item() == 42
    mean = part.double().sum(0)[:, 0] / count                     # and against torch in fp64
    torch.testing.assert_close(outs[1][2].double(), 0.9 * outs[1][5].double() + 0.1 * mean, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('C,nt,B', [(16, 23424, 32), (32, 4100, 8), (64, 2049, 3), (256, 1024, 2), (24, 1500, 2)])
def test_bn_two_level_reduce_matches_one_level(C, nt, B):
    """v2w_bn_reduce_slices + v2w_bn_finalize_slices (two short launches for layers with thousands of partial rows) == v2w_bn_reduce_partials +
    v2w_bn_finalize: same fp64 sums up to their order, same (a, s) and running statistics (models.py:59-70 of the reference in train mode)."""
    from wavthruvec_pytorch_amd import hipops
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(C * 7 + nt)
    part = torch.randn(nt, C, 2, generator=g).abs_().mul_(50).to(dev)
    part[..., 1] += part[..., 0] ** 2 / 100
    count = nt * 224
    gb = torch.randn(B, 2 * C, generator=g).to(dev)
    outs = []
    for two_level in (False, True):
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nb = torch.zeros((), dtype=torch.int64, device=dev)
        a, s = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
        if two_level:
            sl = torch.full((hipops.BN_SLICES * 2 * C,), float('nan'), dtype=torch.float64, device=dev)
            hipops.bn_reduce_finalize_slices(part, nt, count, sl, gb, rm, rv, nb, a, s, momentum=0.1, eps=1e-5)
        else:
            st = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
            hipops.bn_reduce_partials(part, nt, C, count, st)
            hipops.bn_finalize(st, gb, rm, rv, nb, a, s, training=True, momentum=0.1, eps=1e-5)
        outs.append((a, s, rm, rv, nb))
    for x, y in zip(*outs):
        assert torch.isfinite(y.float()).all()
        torch.testing.assert_close(y, x, rtol=1e-6, atol=1e-7)
    ref = part.double().sum(0)                                   # and against torch in fp64
    mean = ref[:, 0] / count
    torch.testing.assert_close(outs[1][2].double(), 0.1 * mean, rtol=1e-6, atol=1e-9)
