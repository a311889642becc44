"""Tensor-level wrappers over the C ABI (one function per entry point of include/vec2wav_hip.h).

torch is plumbing here: it owns device memory and the stream; every arithmetic step runs in the
HIP kernels.  All wrappers require fp32 contiguous CUDA(ROCm) tensors and raise otherwise -
there is deliberately no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _hip

ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_SPLIT, ALGO_BF16 = (_hip.ALGO_AUTO, _hip.ALGO_DIRECT, _hip.ALGO_MFMA, _hip.ALGO_SPLIT,
                                                             _hip.ALGO_BF16)


def _chk(t: Optional[torch.Tensor], name: str, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f'{name} must live on a GPU: the Vec2Wav HIP path has no CPU fallback')
    if t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def fold_conv_weight(v, g, out=None, scratch=None):
    """weight_v (C_out,C_in,k), weight_g (C_out,1,1)|None -> wf [k][C_in][C_out]."""
    _chk(v, 'v'); _chk(g, 'g')
    co, ci, k = v.shape
    if out is None:
        out = torch.empty((k, ci, co), device=v.device, dtype=torch.float32)
    if scratch is None:
        scratch = torch.empty((co,), device=v.device, dtype=torch.float32)
    _hip.check(_hip.load().v2w_wn_fold_conv(v.data_ptr(), _hip.ptr(g), out.data_ptr(), scratch.data_ptr(),
                                            co, ci, k, _stream(v)), 'v2w_wn_fold_conv')
    return out


def fold_convt_weight(v, g, out=None, scratch=None):
    """ConvTranspose1d weight_v (C_in,C_out,k), weight_g (C_in,1,1)|None -> wf [k][C_in][C_out]."""
    _chk(v, 'v'); _chk(g, 'g')
    ci, co, k = v.shape
    if out is None:
        out = torch.empty((k, ci, co), device=v.device, dtype=torch.float32)
    if scratch is None:
        scratch = torch.empty((ci,), device=v.device, dtype=torch.float32)
    _hip.check(_hip.load().v2w_wn_fold_convt(v.data_ptr(), _hip.ptr(g), out.data_ptr(), scratch.data_ptr(),
                                             ci, co, k, _stream(v)), 'v2w_wn_fold_convt')
    return out


def transpose_flip(wf, out=None):
    """wf [k][C_in][C_out] -> dgrad weights [k][C_out][C_in] (taps reversed)."""
    k, ci, co = wf.shape
    if out is None:
        out = torch.empty((k, co, ci), device=wf.device, dtype=torch.float32)
    _hip.check(_hip.load().v2w_wf_transpose_flip(wf.data_ptr(), out.data_ptr(), k, ci, co, _stream(wf)), 'v2w_wf_transpose_flip')
    return out


def gather_transpose(wf, t_start, t_step, n, out=None):
    """out[m][C_out][C_in] = wf[t_start + m*t_step][C_in][C_out], m < n (dgrad weights of one transposed-conv phase)."""
    k, ci, co = wf.shape
    if out is None:
        out = torch.empty((n, co, ci), device=wf.device, dtype=torch.float32)
    _hip.check(_hip.load().v2w_wf_gather_transpose(wf.data_ptr(), out.data_ptr(), k, ci, co, t_start, t_step, n, _stream(wf)),
               'v2w_wf_gather_transpose')
    return out


def pack_mfma(wf, out=None, u=1):
    """wf [k][C_in][C_out] -> the same weights as the MFMA A-fragment stream of that layer (u = 1 conv, stride for convT),
    or None when the layer has no MFMA tile configuration."""
    k, ci, co = wf.shape
    if out is None:
        out = torch.empty((k * ci * co,), device=wf.device, dtype=torch.float32)
    rc = _hip.load().v2w_pack_mfma(wf.data_ptr(), out.data_ptr(), k, ci, co, u, _stream(wf))
    if rc == -2:
        return None
    _hip.check(rc, 'v2w_pack_mfma')
    return out


def pack_mfma_dgrad(wf):
    """wf [k][C_in][C_out] of a conv layer -> the MFMA fragment stream of its input-gradient conv (C_out -> C_in channels, taps reversed),
    or None when that shape has no MFMA tile configuration (then: transpose_flip + the direct kernel)."""
    k, ci, co = wf.shape
    out = torch.empty((k * ci * co,), device=wf.device, dtype=torch.float32)
    rc = _hip.load().v2w_pack_mfma_dgrad(wf.data_ptr(), out.data_ptr(), k, co, ci, _stream(wf))
    if rc == -2:
        return None
    _hip.check(rc, 'v2w_pack_mfma_dgrad')
    return out


def pack_mfma_batch(w4):
    """w4 [n][k][C_in][C_out] contiguous -> the n packed MFMA streams [n][k*C_in*C_out] in one launch."""
    _chk(w4, 'w4')
    n, k, ci, co = w4.shape
    out = torch.empty((n, k * ci * co), device=w4.device, dtype=torch.float32)
    _hip.check(_hip.load().v2w_pack_mfma_batch(w4.data_ptr(), out.data_ptr(), k, ci, co, 1, n, _stream(w4)), 'v2w_pack_mfma_batch')
    return out


def split_supported(c_in, c_out, u=1):
    """True when the split-f16 kernel (ALGO_SPLIT) serves this layer shape."""
    return bool(_hip.load().v2w_split_supported(c_in, c_out, u))


def split_halves(k, c_in, c_out):
    """Elements (f16) of the wps buffer of one layer: two halves per weight + one 2 KiB unit of padding (the kernel's
    stage copies always move whole tap pairs)."""
    return k * c_in * ((c_out + 31) // 32 * 32) * 2 + 1024      # C_out = 16 is zero-padded to one 32-row block


def split_packable(c_in, c_out):
    """True when pack_split / SplitPlan accept the layer shape (the fused C = 32 stage kernel consumes such fragments too)."""
    return bool(_hip.load().v2w_split_packable(c_in, c_out))


def split_units_halves(k, c_in, c_out):
    """f16 elements of the fragment units of one layer WITHOUT the trailing padding unit (slices of a shared stage buffer)."""
    return k * c_in * ((c_out + 31) // 32 * 32) * 2


def pack_split(wf, out=None, sc=None, bf16=False):
    """wf [k][C_in][C_out] -> (wps, sc): the (hi, lo) half-precision MFMA fragments of scale*wf for ALGO_SPLIT and the 4-float
    scale record (sc[0] = 1/scale is the kernel's `winv`)."""
    k, ci, co = wf.shape
    if out is None:
        out = torch.empty((split_halves(k, ci, co),), device=wf.device, dtype=torch.float16)
    if sc is None:
        sc = torch.empty((4,), device=wf.device, dtype=torch.float32)
    fn = _hip.load().v2w_pack_bf16 if bf16 else _hip.load().v2w_pack_split
    _hip.check(fn(wf.data_ptr(), out.data_ptr(), sc.data_ptr(), k, ci, co, _stream(wf)), 'v2w_pack_bf16' if bf16 else 'v2w_pack_split')
    return out, sc


def _conv1d_args(a, x, wf, bias, out, *, k, dil=1, slope=1.0, in_affine=None, res=None, res_affine=None,
                 accumulate=False, out_div=0.0, algo=ALGO_AUTO, wp=None, add=None, mask=None, mask_slope=1.0,
                 in_stride=0, in_phase=0, pad_left=-1, L=None, wps=None, out_slope=0.0, group=None, io_bf16=0, rowsum=None, in_ct=0):
    B, ci, Lx = x.shape
    L = Lx if L is None else L       # strided input: the conv length is Lx / in_stride
    co = out.shape[1]
    a.out_slope = out_slope
    a.rowsum_part = _hip.ptr(rowsum)   # masked f32 MFMA launches: [tiles][C_out][2] per-tile sums of the stored values (bias gradients)
    a.io_bf16 = io_bf16        # ALGO_BF16 only: bit 0 x is bf16, bit 1 out / res / add are bf16 (activation storage of BASELINE configs[2])
    xoff = ooff = 0
    if group is not None:             # (g, C_in per group, C_out per group): this call computes ONE group of a grouped conv
        g, cig, cog = group
        a.in_ct, a.out_ct = ci, co
        xoff, ooff = g * cig * Lx * 4, g * cog * out.shape[2] * 4
        ci, co = cig, cog
    if in_ct:                          # `x` is a C_in-channel slice (a view) of a tensor with in_ct channels per batch item
        a.in_ct = in_ct
    a.in_stride, a.in_phase, a.pad_left = in_stride, in_phase, pad_left
    a.in_ = x.data_ptr() + xoff
    a.in_a, a.in_s = (_hip.ptr(in_affine[0]), _hip.ptr(in_affine[1])) if in_affine is not None else (None, None)
    a.wf = _hip.ptr(wf); a.wp = _hip.ptr(wp); a.bias = _hip.ptr(bias)
    if wps is not None:           # (fragments, scale record) of pack_split
        a.wps, a.winv = wps[0].data_ptr(), wps[1].data_ptr()
    a.res = _hip.ptr(res)
    a.res_a, a.res_s = (_hip.ptr(res_affine[0]), _hip.ptr(res_affine[1])) if res_affine is not None else (None, None)
    add = list(add or [])
    a.add0 = _hip.ptr(add[0]) if len(add) > 0 else None
    a.add1 = _hip.ptr(add[1]) if len(add) > 1 else None
    if mask is not None:          # (mask_src, (mask_a, mask_s) | None)
        a.mask_src = mask[0].data_ptr()
        a.mask_a, a.mask_s = (mask[1][0].data_ptr(), mask[1][1].data_ptr()) if mask[1] is not None else (None, None)
    a.mask_slope = mask_slope
    a.out = out.data_ptr() + ooff
    a.B, a.C_in, a.C_out, a.L, a.k, a.dil = B, ci, co, L, k, dil
    a.slope = slope; a.accumulate = int(accumulate); a.out_div = out_div; a.algo = algo


class SplitKSlab:
    """Caller-owned scratch of the split over C_in (v2w_conv1d_args::splitk_ws, ABI v28): launches too small to fill the chip - inference at
    B = 1 - store per-slice partial sums here and a second kernel adds them.  The library allocates nothing: the OWNER of a slab (a
    Generator, one slab per stream it launches on) passes it to conv1d / conv1d_multi / convt1d as `splitk_ws=`; launches that share a
    slab must be ordered on one stream.  Grow-only, so a warmed-up module never allocates inside a HIP graph capture; a slab that is outgrown is
    kept (`retired`), never freed under a captured graph that writes its partial sums there."""

    def __init__(self):
        self.t = None
        self.retired = []     # superseded slabs stay alive with their owner: a HIP graph or a launch plan recorded earlier holds their address

    def ensure(self, nbytes, device):
        if self.t is None or self.t.device != device or self.t.numel() * 4 < nbytes:
            if self.t is not None:
                self.retired.append(self.t)
            self.t = torch.empty(((nbytes + 3) // 4,), device=device, dtype=torch.float32)
        return self.t


def _attach_slab(a, slab, nbytes, device):
    if slab is not None and nbytes > 0:
        t = slab.ensure(int(nbytes), device)
        a.splitk_ws, a.splitk_ws_bytes = t.data_ptr(), t.numel() * 4


_SPLITK_BYTES = {}      # what a launch's split over C_in needs depends on its sizes and switches alone: asked of the library once per combination


def _splitk_bytes(a, n):
    """Bytes of split-over-C_in scratch the launch of these n problems would use (0: unsplit).  The library's answer depends on EVERY problem of
    a multi-problem launch - tap count and dilation (the widest halo picks the tile variant and the grid), the epilogue kind of each - so the key
    holds them all (keyed by problem 0 alone, two launches that differ in problems 1.. could share a stale byte count)."""
    qs = [a[i] for i in range(n)] if n > 1 else [a]
    key = (n,) + tuple((q.B, q.C_in, q.C_out, q.L, q.k, q.dil, q.algo, q.accumulate, q.in_stride, q.pad_left, q.in_ct, q.out_ct, q.io_bf16,
                        bool(q.res), bool(q.add0), bool(q.add1), bool(q.mask_src), bool(q.in_a), bool(q.rowsum_part), bool(q.wps), bool(q.wp),
                        q.out_div != 0.0, q.out_slope != 0.0) for q in qs)
    v = _SPLITK_BYTES.get(key)
    if v is None:
        v = _SPLITK_BYTES[key] = int(_hip.load().v2w_conv1d_splitk_ws_bytes(a if n > 1 else C.byref(a), n))
    return v


def conv1d(x, wf, bias, out, splitk_ws=None, **kw):
    """Fused [affine] -> leaky_relu -> dilated Conv1d -> +bias [+res] [+= out | + add0 (+ add1)] [/ out_div]; see the header.
    splitk_ws: a SplitKSlab (f32 MFMA path only; without one a small launch simply runs unsplit)."""
    a = _hip.Conv1dArgs()
    _conv1d_args(a, x, wf, bias, out, **kw)
    if splitk_ws is not None:
        _attach_slab(a, splitk_ws, _splitk_bytes(a, 1), x.device)
    _hip.check(_hip.load().v2w_conv1d_fwd(C.byref(a), _stream(x)), 'v2w_conv1d_fwd')
    return out


def conv1d_multi(problems, splitk_ws=None):
    """`problems`: list of (x, wf, bias, out, kwargs) sharing B, C_in, C_out, L.  One launch when the MFMA path takes them
    (heaviest first), otherwise one launch each."""
    n = len(problems)
    if 1 < n <= 4:
        arr = (_hip.Conv1dArgs * n)()
        for a, (x, wf, bias, out, kw) in zip(arr, problems):
            _conv1d_args(a, x, wf, bias, out, **kw)
        if splitk_ws is not None:
            _attach_slab(arr[0], splitk_ws, _splitk_bytes(arr, n), problems[0][0].device)
        rc = _hip.load().v2w_conv1d_fwd_multi(arr, n, _stream(problems[0][0]))
        if rc == 0:
            return
        if rc != -2:
            _hip.check(rc, 'v2w_conv1d_fwd_multi')
    for x, wf, bias, out, kw in problems:
        conv1d(x, wf, bias, out, splitk_ws=splitk_ws, **kw)


def convt1d(x, wf, bias, out, *, k, u, slope=1.0, algo=ALGO_AUTO, wp=None, stats_part=None, splitk_ws=None):
    """Fused leaky_relu -> ConvTranspose1d(k, stride u, padding (k-u)//2) -> +bias."""
    B, ci, L = x.shape
    a = _hip.ConvT1dArgs()
    a.in_ = x.data_ptr(); a.wf = _hip.ptr(wf); a.wp = _hip.ptr(wp); a.bias = _hip.ptr(bias); a.out = out.data_ptr()
    a.stats_part = _hip.ptr(stats_part)
    a.B, a.C_in, a.C_out, a.L, a.k, a.u = B, ci, out.shape[1], L, k, u
    a.slope = slope; a.algo = algo
    if splitk_ws is not None:
        _attach_slab(a, splitk_ws, _hip.load().v2w_convt1d_splitk_ws_bytes(C.byref(a)), x.device)
    _hip.check(_hip.load().v2w_convt1d_fwd(C.byref(a), _stream(x)), 'v2w_convt1d_fwd')
    return out


def pack_bf16_convt(wf, u, out=None):
    """ConvTranspose1d weights wf [k][C_in][C_out] -> bf16 fragments of the virtual 3-tap conv v2w_convt1d_bf16_fwd runs, or None when
    the shape is not served."""
    k, ci, co = wf.shape
    nbytes = int(_hip.load().v2w_pack_bf16_convt_bytes(k, ci, co, u))
    if nbytes == 0:
        return None
    if out is None or out.numel() * out.element_size() < nbytes:
        out = torch.empty((nbytes // 2,), device=wf.device, dtype=torch.bfloat16)
    _hip.check(_hip.load().v2w_pack_bf16_convt(wf.data_ptr(), out.data_ptr(), k, ci, co, u, _stream(wf)), 'v2w_pack_bf16_convt')
    return out


def _convt_bf16_args(x, wps, bias, out, k, u, slope, stats_part, io_bf16=0):
    B, ci, L = x.shape
    a = _hip.ConvT1dArgs()
    a.in_ = x.data_ptr(); a.wf = None; a.wp = _hip.ptr(wps); a.bias = _hip.ptr(bias); a.out = _hip.ptr(out)
    a.stats_part = _hip.ptr(stats_part)
    a.B, a.C_in, a.C_out, a.L, a.k, a.u = B, ci, out.shape[1], L, k, u
    a.slope = slope; a.algo = ALGO_BF16; a.io_bf16 = io_bf16
    return a


def convt_bf16_stats_tiles(x, out, k, u, io_bf16=0):
    """Rows of `stats_part` the bf16 transposed conv fills for exactly this call (tensors, io_bf16: the kernel and with it the tile
    width follow them); 0: shape not served."""
    n = _hip.load().v2w_convt1d_bf16_tiles(C.byref(_convt_bf16_args(x, None, None, out, k, u, 1.0, None, io_bf16)))
    return n if n > 0 else 0


def convt1d_bf16(x, wps, bias, out, *, k, u, slope=1.0, stats_part=None, io_bf16=0):
    """Fused leaky_relu -> ConvTranspose1d(k, stride u, padding (k-u)//2) -> +bias on the bf16 matrix pipe (fp32 accumulate).
    io_bf16 = 3: x and out are bf16 tensors."""
    _hip.check(_hip.load().v2w_convt1d_bf16_fwd(C.byref(_convt_bf16_args(x, wps, bias, out, k, u, slope, stats_part, io_bf16)), _stream(x)),
               'v2w_convt1d_bf16_fwd')
    return out


def cond_gamma_beta(spk, noise, fc_w: Sequence, fc_b: Sequence, sn_w: Sequence, sn_b: Sequence,
                    sn_u: Sequence, sn_v: Sequence, gb: Sequence, z_ws, sigma_ws, training: bool):
    """All stages' [gamma|beta] in one call; sn_u / sn_v are updated in place when training."""
    n = len(fc_w)
    if n > _hip.V2W_MAX_STAGES:
        raise ValueError(f'at most {_hip.V2W_MAX_STAGES} upsample stages are supported')
    a = _hip.CondArgs()
    a.spk = spk.data_ptr(); a.noise = noise.data_ptr()
    for i in range(n):
        a.fc_w[i] = fc_w[i].data_ptr(); a.fc_b[i] = fc_b[i].data_ptr()
        a.sn_w[i] = sn_w[i].data_ptr(); a.sn_b[i] = sn_b[i].data_ptr()
        a.sn_u[i] = sn_u[i].data_ptr(); a.sn_v[i] = sn_v[i].data_ptr()
        a.gb[i] = gb[i].data_ptr()
        a.C[i] = sn_w[i].shape[0] // 2
    a.z_ws = z_ws.data_ptr(); a.sigma_ws = sigma_ws.data_ptr()
    a.n_stages = n; a.B = spk.shape[0]; a.spk_dim = spk.shape[1]; a.noise_dim = noise.shape[1]
    a.training = int(training)
    _hip.check(_hip.load().v2w_cond_gamma_beta(C.byref(a), _stream(spk)), 'v2w_cond_gamma_beta')


def _cond_args(a, spk, noise, fc_w, fc_b, sn_w, sn_b, sn_u, sn_v, sigma_ws):
    n = len(sn_w)
    if n > _hip.V2W_MAX_STAGES:
        raise ValueError(f'at most {_hip.V2W_MAX_STAGES} upsample stages are supported')
    a.spk = _hip.ptr(spk); a.noise = _hip.ptr(noise)
    for i in range(n):
        a.fc_w[i] = _hip.ptr(fc_w[i]) if fc_w is not None else None
        a.fc_b[i] = _hip.ptr(fc_b[i]) if fc_b is not None else None
        a.sn_w[i] = sn_w[i].data_ptr(); a.sn_b[i] = _hip.ptr(sn_b[i]) if sn_b is not None else None
        a.sn_u[i] = _hip.ptr(sn_u[i]) if sn_u is not None else None
        a.sn_v[i] = _hip.ptr(sn_v[i]) if sn_v is not None else None
        a.C[i] = sn_w[i].shape[0] // 2
    a.sigma_ws = sigma_ws.data_ptr()
    a.n_stages = n
    if spk is not None:
        a.B = spk.shape[0]; a.spk_dim = spk.shape[1]; a.noise_dim = noise.shape[1]


def cond_sigma(sn_w: Sequence, sn_u: Sequence, sn_v: Sequence, sigma_ws, training: bool = False):
    """sigma_ws[i] = u_i . (W_i v_i) of every stage (eval: no power iteration - a function of the parameters alone)."""
    a = _hip.CondArgs()
    _cond_args(a, None, None, None, None, sn_w, None, sn_u, sn_v, sigma_ws)
    a.training = int(training)
    _hip.check(_hip.load().v2w_cond_sigma(C.byref(a), _stream(sigma_ws)), 'v2w_cond_sigma')


def cond_affine_eval(spk, noise, fc_w: Sequence, fc_b: Sequence, sn_w: Sequence, sn_b: Sequence, sigma_ws,
                     running_mean: Sequence, running_var: Sequence, eps: Sequence, a_out: Sequence, s_out: Sequence):
    """Eval mode: the folded per-sample affine (a, s) of every stage's Conditional BatchNorm from (spk, noise) in ONE launch
    (sigma_ws from cond_sigma for the current parameters)."""
    e = _hip.CondEvalArgs()
    _cond_args(e.c, spk, noise, fc_w, fc_b, sn_w, sn_b, None, None, sigma_ws)
    for i in range(len(sn_w)):
        e.running_mean[i] = running_mean[i].data_ptr(); e.running_var[i] = running_var[i].data_ptr()
        e.a_out[i] = a_out[i].data_ptr(); e.s_out[i] = s_out[i].data_ptr()
        e.eps[i] = eps[i]
    _hip.check(_hip.load().v2w_cond_affine_eval(C.byref(e), _stream(spk)), 'v2w_cond_affine_eval')


def bn_stats(x, stats, partial_ws):
    """x (B,C,L) -> stats[2C+1] fp64 = [sum | sumsq | count]."""
    B, Cc, L = x.shape
    _hip.check(_hip.load().v2w_bn_stats(x.data_ptr(), stats.data_ptr(), partial_ws.data_ptr(), B, Cc, L, _stream(x)),
               'v2w_bn_stats')
    return stats


def bn_finalize(stats, gb, running_mean, running_var, num_batches_tracked, a_out, s_out, *, training: bool,
                momentum=0.1, eps=1e-5):
    B, C2 = gb.shape
    _hip.check(_hip.load().v2w_bn_finalize(_hip.ptr(stats), gb.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                                           _hip.ptr(num_batches_tracked), a_out.data_ptr(), s_out.data_ptr(),
                                           B, C2 // 2, int(training), momentum, eps, _stream(gb)), 'v2w_bn_finalize')


BN_SLICES = 128


def bn_reduce_finalize_slices(part, ntiles, count, slices, gb, running_mean, running_var, num_batches_tracked, a_out, s_out, *,
                              momentum=0.1, eps=1e-5):
    """Train-mode BatchNorm statistics of a layer whose producer left `ntiles` rows of partial sums, and their fold into (a, s), as two short
    launches: slices of whole rows -> fp64 sums per slice (`slices`: >= BN_SLICES * 2 C doubles), then the finalize kernel adds the slices."""
    B, C2 = gb.shape
    Cc = C2 // 2
    lib, st = _hip.load(), _stream(part)
    _hip.check(lib.v2w_bn_reduce_slices(part.data_ptr(), ntiles, Cc, slices.data_ptr(), BN_SLICES, st), 'v2w_bn_reduce_slices')
    _hip.check(lib.v2w_bn_finalize_slices(slices.data_ptr(), BN_SLICES, float(count), gb.data_ptr(), running_mean.data_ptr(),
                                          running_var.data_ptr(), _hip.ptr(num_batches_tracked), a_out.data_ptr(), s_out.data_ptr(),
                                          B, Cc, momentum, eps, st), 'v2w_bn_finalize_slices')


def affine_apply(x, a, s, out):
    B, Cc, L = x.shape
    _hip.check(_hip.load().v2w_affine_apply(x.data_ptr(), a.data_ptr(), s.data_ptr(), out.data_ptr(), B, Cc, L, _stream(x)),
               'v2w_affine_apply')
    return out


def conv_post_tanh(x, wf, bias, out, *, k, slope):
    B, ci, L = x.shape
    if x.dtype == torch.bfloat16:        # bf16 activation storage: the tail reads bf16, computes and writes fp32
        _hip.check(_hip.load().v2w_conv_post_tanh_bf16in(x.data_ptr(), wf.data_ptr(), _hip.ptr(bias), out.data_ptr(),
                                                         B, ci, L, k, slope, _stream(x)), 'v2w_conv_post_tanh_bf16in')
        return out
    _hip.check(_hip.load().v2w_conv_post_tanh(x.data_ptr(), wf.data_ptr(), _hip.ptr(bias), out.data_ptr(),
                                              B, ci, L, k, slope, _stream(x)), 'v2w_conv_post_tanh')
    return out


def conv_tile_config(B, c_in, c_out, L, k, dil=1, u=1, prefix=False):
    """Name of the conv_tile_kernel instantiation the MFMA path picks for this problem, or None (direct kernel).  The instantiation's
    last template argument (VEC: vector staging) depends on the alignment and stride of the actual input, which a shape query cannot
    know: the name carries the value an aligned, unit-stride tensor of this length gets (L % 4 == 0); `prefix=True` returns the name
    up to that argument, for `startswith` matching of trace names."""
    cfg = (C.c_int32 * 10)()
    if u == 1:
        a = _hip.Conv1dArgs(); a.B, a.C_in, a.C_out, a.L, a.k, a.dil = B, c_in, c_out, L, k, dil
        a.pad_left = -1
        rc = _hip.load().v2w_conv1d_tile_config(C.byref(a), cfg)
    else:
        a = _hip.ConvT1dArgs(); a.B, a.C_in, a.C_out, a.L, a.k, a.u = B, c_in, c_out, L, k, u
        rc = _hip.load().v2w_convt1d_tile_config(C.byref(a), cfg)
    if rc != 0:
        return None
    # forward instantiation (EPI = 0: no optional epilogue)
    pre = 'conv_tile_kernel<' + ', '.join(str(v) for v in cfg[:9]) + ', 0, '
    return pre if prefix else pre + ('true>' if L % 4 == 0 else 'false>')


def conv_bf16_config(B, nprob, c_in, c_out, L, k, dil=1, u=1, io_bf16=3):
    """Name of the conv_bf16_kernel instantiation a V2W_ALGO_BF16 launch of `nprob` such problems runs (aligned tensors), or None."""
    cfg = (C.c_int32 * 10)()
    if u == 1:
        arr = (_hip.Conv1dArgs * nprob)()
        for a in arr:
            a.B, a.C_in, a.C_out, a.L, a.k, a.dil = B, c_in, c_out, L, k, dil
            a.pad_left = -1
            a.io_bf16 = io_bf16
        rc = _hip.load().v2w_conv1d_bf16_config(arr, nprob, cfg)
    else:
        a = _hip.ConvT1dArgs(); a.B, a.C_in, a.C_out, a.L, a.k, a.u = B, c_in, c_out, L, k, u
        a.io_bf16 = io_bf16
        a.slope = 0.1              # (the generator's upsamplers; the resident-tile kernel rebuilds nothing from it but checks 0 < slope <= 1)
        rc = _hip.load().v2w_convt1d_bf16_config(C.byref(a), cfg)
    if rc != 0:
        return None
    v = list(cfg)
    if v[5] == 102:         # the resident-tile transposed conv (v2w_convt_bf16_res.hip)
        return 'convt_bf16_res_kernel<' + ', '.join(str(x) for x in v[:5]) + f', {v[9]}>'      # (..., UP, U)
    tf = lambda b: 'true' if b else 'false'
    return 'conv_bf16_kernel<' + ', '.join(str(x) for x in v[:6]) + f', {tf(v[6])}, {tf(v[7])}, {v[8]}, {tf(v[9])}>'


def conv_rowsum_tiles(B, nprob, c_in, c_out, L, k, dil=1):
    """Rows of the `rowsum` array a masked f32 MFMA launch of `nprob` problems of this shape fills per problem (0: no tile configuration)."""
    cfg = (C.c_int32 * 10)()
    a = _hip.Conv1dArgs(); a.B, a.C_in, a.C_out, a.L, a.k, a.dil = B * nprob, c_in, c_out, L, k, dil
    a.pad_left = -1
    if _hip.load().v2w_conv1d_tile_config(C.byref(a), cfg) != 0:
        return 0
    return cfg[9] // nprob


def convt_stats_tiles(B, c_in, c_out, L, k, u):
    """Rows of the `stats_part` array the MFMA transposed conv fills for this problem (0: direct kernel, no fused stats)."""
    cfg = (C.c_int32 * 10)()
    a = _hip.ConvT1dArgs(); a.B, a.C_in, a.C_out, a.L, a.k, a.u = B, c_in, c_out, L, k, u
    return int(cfg[9]) if _hip.load().v2w_convt1d_tile_config(C.byref(a), cfg) == 0 else 0


def bn_reduce_partials(part, ntiles, Cc, count, stats):
    _hip.check(_hip.load().v2w_bn_reduce_partials(part.data_ptr(), ntiles, Cc, float(count), stats.data_ptr(), _stream(part)),
               'v2w_bn_reduce_partials')
    return stats


def bn_reduce_finalize(part, ntiles, count, stats, gb, running_mean, running_var, num_batches_tracked, a_out, s_out, *, momentum=0.1, eps=1e-5):
    """bn_reduce_partials + bn_finalize(training=True) in one launch (v2w_bn_reduce_finalize): same values, bit for bit."""
    B, C2 = gb.shape
    _hip.check(_hip.load().v2w_bn_reduce_finalize(part.data_ptr(), ntiles, float(count), gb.data_ptr(), running_mean.data_ptr(),
                                                  running_var.data_ptr(), _hip.ptr(num_batches_tracked), stats.data_ptr(), a_out.data_ptr(),
                                                  s_out.data_ptr(), B, C2 // 2, momentum, eps, _stream(gb)), 'v2w_bn_reduce_finalize')
    return a_out, s_out


class FoldPlan:
    """Device-resident descriptor table for v2w_fold_pack_batch: every MFMA layer folded + packed in two launches."""

    def __init__(self, layers, device):
        """layers: list of (v, g|None, wp, c_in, c_out, k, u, transposed[, wf|None[, wpd|None]]) with tensors already on `device`; wf: the plain
        layout [k][C_in][C_out] as well, wpd: the fragment stream of the layer's input-gradient conv as well (training forwards)."""
        n = len(layers)
        self.n = n
        layers = [tuple(l) + (None,) * (10 - len(l)) for l in layers]
        rows = [(l[3] if l[7] else l[4]) for l in layers]
        self.scale = torch.empty((sum(rows),), device=device, dtype=torch.float32)
        descs = (_hip.FoldDesc * n)()
        off = 0
        for d, (v, g, wp, ci, co, k, u, tr, wf, wpd), r in zip(descs, layers, rows):
            d.v = v.data_ptr(); d.g = _hip.ptr(g); d.wp = wp.data_ptr()
            d.wf = _hip.ptr(wf); d.wpd = _hip.ptr(wpd)
            d.scale = self.scale.data_ptr() + 4 * off
            d.c_in, d.c_out, d.k, d.u, d.transposed = ci, co, k, u, int(tr)
            off += r
        starts = (C.c_int32 * (2 * (n + 1)))()
        lds = _hip.load().v2w_fold_plan(descs, n, starts)
        _hip.check(0 if lds > 0 else (lds or -1), 'v2w_fold_plan')
        self.lds = lds
        self.nblk_scale, self.nblk_pack = starts[n], starts[2 * n + 1]
        self.descs_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
        self.starts_dev = torch.tensor(list(starts), dtype=torch.int32, device=device)
        self.key = tuple((v.data_ptr(), 0 if g is None else g.data_ptr(), wp.data_ptr()) for (v, g, wp, *_r) in layers)

    def run(self):
        _hip.check(_hip.load().v2w_fold_pack_batch(self.descs_dev.data_ptr(), self.starts_dev.data_ptr(), self.n,
                                                   self.nblk_scale, self.nblk_pack, self.lds,
                                                   torch.cuda.current_stream(self.scale.device).cuda_stream),
                   'v2w_fold_pack_batch')


class SplitPlan:
    """Device-resident descriptor table for v2w_split_pack_batch: every split-f16 layer folded (weight norm) and packed into
    its (hi, lo) fragment stream in three launches."""

    def __init__(self, layers, device, bf16=False):
        """layers: list of (v (C_out, C_in, k), g|None, wps, sc) with tensors already on `device`; bf16: fragments for ALGO_BF16."""
        n = len(layers)
        self.n = n
        self.bf16 = bool(bf16)
        self.rowscale = torch.empty((sum(v.shape[0] for v, *_ in layers),), device=device, dtype=torch.float32)
        descs = (_hip.SplitDesc * n)()
        starts = [0] * (2 * (n + 1))
        off = blocks = 0
        for i, (d, (v, g, wps, sc)) in enumerate(zip(descs, layers)):
            co, ci, k = v.shape
            d.v = v.data_ptr(); d.g = _hip.ptr(g); d.wps = wps.data_ptr(); d.sc = sc.data_ptr()
            d.rowscale = self.rowscale.data_ptr() + 4 * off
            d.c_in, d.c_out, d.k, d.mode = ci, co, k, int(bf16)
            starts[i], starts[n + 1 + i] = off, blocks
            off += co
            blocks += ((co + 31) // 32) * (ci // 16)
        starts[n], starts[2 * n + 1] = off, blocks
        self.nblk_rows, self.nblk_pack = off, blocks
        self.k_max = max(v.shape[2] for v, *_ in layers)
        self.descs_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
        self.starts_dev = torch.tensor(starts, dtype=torch.int32, device=device)
        self.key = tuple((v.data_ptr(), 0 if g is None else g.data_ptr(), w.data_ptr()) for (v, g, w, _s) in layers)

    def run(self):
        _hip.check(_hip.load().v2w_split_pack_batch(self.descs_dev.data_ptr(), self.starts_dev.data_ptr(), self.n,
                                                    self.nblk_rows, self.nblk_pack, self.k_max, int(self.bf16),
                                                    torch.cuda.current_stream(self.rowscale.device).cuda_stream),
                   'v2w_split_pack_batch')


def resblock_pair_multi(problems):
    """`problems`: list of dicts(x, in_affine, wp1, b1, wp2, b2, out, k, dil1, dil2, res_mode, slope, add, out_div) sharing B, C, L.
    Returns False (nothing launched) when the fused pair kernel does not take the shape."""
    n = len(problems)
    arr = (_hip.PairArgs * n)()
    for a, q in zip(arr, problems):
        x = q['x']
        B, Cc, L = x.shape
        a.in_ = x.data_ptr()
        aff = q.get('in_affine')
        a.in_a, a.in_s = (aff[0].data_ptr(), aff[1].data_ptr()) if aff is not None else (None, None)
        a.wp1 = q['wp1'].data_ptr(); a.bias1 = _hip.ptr(q['b1']); a.wp2 = q['wp2'].data_ptr(); a.bias2 = _hip.ptr(q['b2'])
        add = list(q.get('add') or [])
        a.add0 = _hip.ptr(add[0]) if len(add) > 0 else None
        a.add1 = _hip.ptr(add[1]) if len(add) > 1 else None
        a.out = q['out'].data_ptr()
        a.B, a.C, a.L, a.k, a.dil1, a.dil2 = B, Cc, L, q['k'], q['dil1'], q['dil2']
        a.res_mode = q['res_mode']; a.slope = q['slope']; a.out_div = q.get('out_div', 0.0)
    rc = _hip.load().v2w_resblock_pair_fwd(arr, n, _stream(problems[0]['x']))
    if rc == -2:
        return False
    _hip.check(rc, 'v2w_resblock_pair_fwd')
    return True


def resblock1_pairs_ok(B, Cc, L, ks, dil1s, dil2s, *, slope) -> bool:
    """Shape query: would resblock1_pairs_bf16 take these problems (kernel sizes `ks`, dilations of the first / second conv of the pair)?"""
    a = _hip.StageSplitArgs()
    a.rb1 = 1
    for j, (k, d1, d2) in enumerate(zip(ks, dil1s, dil2s)):
        a.k[j], a.dil1[j], a.dil2[j] = k, d1, d2
    a.nk, a.B, a.C, a.L = len(ks), B, Cc, L
    a.slope, a.out_div, a.bf16, a.io_bf16 = slope, 0.0, 1, 3
    return _hip.load().v2w_resblock2_stage_split_config(C.byref(a)) == 0


def resblock2_stage_split_ok(B, Cc, L, ks, dil1s, dil2s, *, slope, bf16=True, io_bf16=3) -> bool:
    """Shape query: would resblock2_stage_split run this stage (nk branches of kernel sizes `ks`, dilations `dil1s` / `dil2s`) as one
    kernel on aligned tensors?  Asked of the library (v2w_resblock2_stage_split_config), nothing is launched."""
    a = _hip.StageSplitArgs()
    for j, (k, d1, d2) in enumerate(zip(ks, dil1s, dil2s)):
        a.k[j], a.dil1[j], a.dil2[j] = k, d1, d2
    a.nk, a.B, a.C, a.L = len(ks), B, Cc, L
    a.slope, a.out_div, a.bf16, a.io_bf16 = slope, float(len(ks)), int(bf16), io_bf16
    return _hip.load().v2w_resblock2_stage_split_config(C.byref(a)) == 0


def resblock2_stage_up_tiles(B, Cc, L, ks, dil1s, dil2s, *, slope, up_k, up_u, up_slope) -> int:
    """Rows of the `stats_part` array the stage kernel WITH the next stage's upsampler fused behind it fills (resblock2_stage_split(up=...)),
    or 0 when the library does not run this stage fused (shape query, nothing is launched)."""
    a = _hip.StageSplitArgs()
    for j, (k, d1, d2) in enumerate(zip(ks, dil1s, dil2s)):
        a.k[j], a.dil1[j], a.dil2[j] = k, d1, d2
    a.nk, a.B, a.C, a.L = len(ks), B, Cc, L
    a.slope, a.out_div, a.bf16, a.io_bf16 = slope, float(len(ks)), 1, 3
    a.up_k, a.up_u, a.up_slope = up_k, up_u, up_slope
    n = _hip.load().v2w_resblock2_stage_up_tiles(C.byref(a))
    return n if n > 0 else 0


def resblock1_pairs_bf16(ins, in_affine, branches, outs, *, slope, out_div=0.0, add=None):
    """ResBlock1 on bf16 tensors (v2w_stage_split_args::rb1): len(branches) independent problems in one launch, problem p computing
    outs[p] = (x_p + conv2_p(lrelu(conv1_p(lrelu x_p) + b1_p)) + b2_p [+ add[0] + add[1]]) / out_div with x_p = a * ins[p] + s.  `branches`:
    dicts(wps1, b1, wps2, b2, k, dil1, dil2) as for resblock2_stage_split; `add`: up to two bf16 tensors the LAST problem adds before the
    division.  Returns False when the shape is not taken."""
    B, Cc, L = ins[0].shape
    a = _hip.StageSplitArgs()
    a.rb1 = 1
    a.in_a, a.in_s = (in_affine[0].data_ptr(), in_affine[1].data_ptr()) if in_affine is not None else (None, None)
    for j, q in enumerate(branches):
        a.wps1[j], a.sc1[j], a.bias1[j] = q['wps1'][0].data_ptr(), q['wps1'][1].data_ptr(), _hip.ptr(q['b1'])
        a.wps2[j], a.sc2[j], a.bias2[j] = q['wps2'][0].data_ptr(), q['wps2'][1].data_ptr(), _hip.ptr(q['b2'])
        a.k[j], a.dil1[j], a.dil2[j] = q['k'], q['dil1'], q['dil2']
        a.in_b[j] = ins[j].data_ptr(); a.out_b[j] = outs[j].data_ptr()
    add = list(add or [])
    a.add0 = _hip.ptr(add[0]) if len(add) > 0 else None
    a.add1 = _hip.ptr(add[1]) if len(add) > 1 else None
    a.nk, a.B, a.C, a.L = len(branches), B, Cc, L
    a.slope, a.out_div, a.bf16, a.io_bf16 = slope, out_div, 1, 3
    rc = _hip.load().v2w_resblock2_stage_split_fwd(C.byref(a), _stream(ins[0]))
    if rc == -2:
        return False
    _hip.check(rc, 'v2w_resblock2_stage_split_fwd (rb1)')
    return True


def resblock2_stage_split(x, in_affine, branches, out, *, slope, out_div, bf16=False, io_bf16=0, post=None, up=None):
    """Split-operand (f16x3 / bf16) form of resblock2_stage for C == 32.  `branches`: list of dicts(wps1, b1, wps2, b2, k, dil1, dil2)
    with wps* = (fragments, scale record) of pack_split / SplitPlan.  Returns False when the shape is not taken.
    up = (wps of pack_bf16_convt, bias, out (B, C / 2, u L) bf16, stats_part | None, k, u, slope): the NEXT stage's upsampler run on the
    stage's output inside the same kernel (`out` may be None: it is not written)."""
    B, Cc, L = x.shape
    a = _hip.StageSplitArgs()
    if up is not None:
        a.up_wps, a.up_bias, a.up_out, a.up_stats_part = up[0].data_ptr(), _hip.ptr(up[1]), up[2].data_ptr(), _hip.ptr(up[3])
        a.up_k, a.up_u, a.up_slope = up[4], up[5], up[6]
    a.in_ = x.data_ptr()
    a.in_a, a.in_s = (in_affine[0].data_ptr(), in_affine[1].data_ptr()) if in_affine is not None else (None, None)
    for j, q in enumerate(branches):
        a.wps1[j], a.sc1[j], a.bias1[j] = q['wps1'][0].data_ptr(), q['wps1'][1].data_ptr(), _hip.ptr(q['b1'])
        a.wps2[j], a.sc2[j], a.bias2[j] = q['wps2'][0].data_ptr(), q['wps2'][1].data_ptr(), _hip.ptr(q['b2'])
        a.k[j], a.dil1[j], a.dil2[j] = q['k'], q['dil1'], q['dil2']
    a.out = _hip.ptr(out)
    a.nk, a.B, a.C, a.L = len(branches), B, Cc, L
    a.slope, a.out_div, a.bf16 = slope, out_div, int(bf16)
    a.io_bf16 = io_bf16
    if post is not None:      # (wf [k][C][1], bias | None, y (B, 1, L) fp32, k, slope): the generator's tail fused behind the C = 16 stage
        a.post_w, a.post_b, a.post_out = post[0].data_ptr(), _hip.ptr(post[1]), post[2].data_ptr()
        a.post_k, a.post_slope = post[3], post[4]
    rc = _hip.load().v2w_resblock2_stage_split_fwd(C.byref(a), _stream(x))
    if rc == -2:
        return False
    _hip.check(rc, 'v2w_resblock2_stage_split_fwd')
    return True


def branch_convs_bf16(mode, ins, in_affine, wps, biases, outs, ks, dils, *, slope, out_div=0.0):
    """The first (mode 0) or second (mode 1) convs of the residual branches of a wide ResBlock2 stage on bf16 tensors in ONE launch
    (v2w_branch_convs_bf16_fwd).  mode 0: ins = [x], outs = [t1_j]; mode 1: ins = [t1_j], outs = [out].  False: shape not served."""
    a = _hip.BranchConvsArgs()
    n = len(wps)
    x = ins[0]
    for t in list(ins) + list(outs):
        if t.dtype != torch.bfloat16 or not t.is_contiguous() or t.shape != x.shape:
            raise ValueError('branch_convs_bf16: contiguous bf16 (B, C, L) tensors of one shape')
    B, Cc, L = x.shape
    for j in range(n):
        a.wps[j] = wps[j].data_ptr(); a.bias[j] = _hip.ptr(biases[j]); a.k[j] = ks[j]; a.dil[j] = dils[j]
    for j, t in enumerate(ins):
        a.in_[j] = t.data_ptr()
    for j, t in enumerate(outs):
        a.out[j] = t.data_ptr()
    a.in_a, a.in_s = (_hip.ptr(in_affine[0]), _hip.ptr(in_affine[1])) if in_affine is not None else (None, None)
    a.nbr, a.mode, a.B, a.C, a.L = n, mode, B, Cc, L
    a.slope, a.out_div = slope, out_div
    rc = _hip.load().v2w_branch_convs_bf16_fwd(C.byref(a), _stream(x))
    if rc == -2:
        return False
    _hip.check(rc, 'v2w_branch_convs_bf16_fwd')
    return True


def resblock2_stage(x, in_affine, branches, out, *, slope, out_div, post=None, bwd=None):
    """Whole ResBlock2 residual section of a narrow stage in one kernel.  `branches`: list of dicts(wp1, b1, wp2, b2, k, dil1, dil2).
    post = (wf [k][C][1], bias | None, y (B, 1, L) fp32, k, slope): the generator's tail fused behind the 16-channel stage - `out` may be None,
    it is not written.  bwd = (t1s [nk], dt1s [nk], xr, (a, s) | None, mask_slope[, rowsums [nk]]): the section's input gradient instead (v2w_stage_args::bwd_*:
    x = dL/d(out), in_affine = (1 / nk, 0), wp1 / wp2 the transposed streams of conv2 / conv1, slope 1).
    Returns False (nothing launched) when the shape is not taken."""
    B, Cc, L = x.shape
    a = _hip.StageArgs()
    a.in_ = x.data_ptr()
    a.in_a, a.in_s = (in_affine[0].data_ptr(), in_affine[1].data_ptr()) if in_affine is not None else (None, None)
    for j, br in enumerate(branches):
        a.wp1[j] = br['wp1'].data_ptr(); a.bias1[j] = _hip.ptr(br['b1'])
        a.wp2[j] = br['wp2'].data_ptr(); a.bias2[j] = _hip.ptr(br['b2'])
        a.k[j], a.dil1[j], a.dil2[j] = br['k'], br['dil1'], br['dil2']
    a.out = _hip.ptr(out)
    a.nk, a.B, a.C, a.L = len(branches), B, Cc, L
    a.slope = slope; a.out_div = out_div
    if post is not None:
        a.post_w, a.post_b, a.post_out = post[0].data_ptr(), _hip.ptr(post[1]), post[2].data_ptr()
        a.post_k, a.post_slope = post[3], post[4]
    if bwd is not None:
        t1s, dt1s, xr, xaff, mslope = bwd[:5]
        rowsums = bwd[5] if len(bwd) > 5 else None
        for j in range(len(branches)):
            a.bwd_mask1[j] = t1s[j].data_ptr(); a.bwd_mid[j] = dt1s[j].data_ptr()
            a.bwd_rowsum[j] = _hip.ptr(rowsums[j]) if rowsums is not None else None
        a.bwd_mask2 = xr.data_ptr()
        a.bwd_mask2_a, a.bwd_mask2_s = (xaff[0].data_ptr(), xaff[1].data_ptr()) if xaff is not None else (None, None)
        a.bwd_slope = mslope
    rc = _hip.load().v2w_resblock2_stage_fwd(C.byref(a), _stream(x))
    if rc == -2:
        return False
    _hip.check(rc, 'v2w_resblock2_stage_fwd')
    return True


def resblock2_stage_bwd_rows(B, Cc, L, ks, dil1s, dil2s) -> int:
    """Rows of [C][2] floats per branch the input-gradient form of `resblock2_stage` writes into its `rowsums` buffers (0: shape not taken).
    dil1s / dil2s as handed to the kernel (conv2's dilations first)."""
    a = _hip.StageArgs()
    a.nk, a.B, a.C, a.L = len(ks), B, Cc, L
    for j in range(len(ks)):
        a.k[j], a.dil1[j], a.dil2[j] = ks[j], dil1s[j], dil2s[j]
    return _hip.load().v2w_resblock2_stage_bwd_rows(C.byref(a))


def resblock2_stage_small(x, in_affine, branches, out, *, slope, out_div):
    """The same section for an 8-channel stage (v2w_resblock2_stage_small_fwd: fp32 FMAs, x read once, t1_j in LDS).  `branches`: list of
    dicts(wf1, b1, wf2, b2, k, dil1, dil2) with the FOLDED weights [k][C][C].  Returns False when the shape is not taken."""
    B, Cc, L = x.shape
    a = _hip.StageArgs()
    a.in_ = x.data_ptr()
    a.in_a, a.in_s = (in_affine[0].data_ptr(), in_affine[1].data_ptr()) if in_affine is not None else (None, None)
    for j, br in enumerate(branches):
        a.wp1[j] = br['wf1'].data_ptr(); a.bias1[j] = _hip.ptr(br['b1'])
        a.wp2[j] = br['wf2'].data_ptr(); a.bias2[j] = _hip.ptr(br['b2'])
        a.k[j], a.dil1[j], a.dil2[j] = br['k'], br['dil1'], br['dil2']
    a.out = out.data_ptr()
    a.nk, a.B, a.C, a.L = len(branches), B, Cc, L
    a.slope = slope; a.out_div = out_div
    rc = _hip.load().v2w_resblock2_stage_small_fwd(C.byref(a), _stream(x))
    if rc == -2:
        return False
    _hip.check(rc, 'v2w_resblock2_stage_small_fwd')
    return True


def wgrad(x, dy, *, k, dil=1, u=1, slope=1.0, x_affine=None, out=None):
    """Weight gradient dwf [k][C_in][C_out] of a fused lrelu -> Conv1d (u = 1) or lrelu -> ConvTranspose1d (stride u).
    x (B, C_in, Lq) is the forward conv's input (before the affine / activation), dy (B, C_out, u*Lq) the output gradient."""
    B, ci, Lq = x.shape
    co = dy.shape[1]
    lib = _hip.load()
    ns = lib.v2w_wgrad_slabs(B, ci, co, Lq)
    if ns == 0:
        raise _hip.HipLibraryError(f'v2w_wgrad: no configuration for C_in={ci}, C_out={co}')
    if out is None:
        out = torch.empty((k, ci, co), device=x.device, dtype=torch.float32)
    slab = torch.empty((ns * k * ci * co,), device=x.device, dtype=torch.float32)
    xa, xs = (x_affine[0].data_ptr(), x_affine[1].data_ptr()) if x_affine is not None else (None, None)
    _hip.check(lib.v2w_wgrad(x.data_ptr(), xa, xs, dy.data_ptr(), out.data_ptr(), slab.data_ptr(), B, ci, co, Lq, k, dil, u, slope,
                             _stream(x)), 'v2w_wgrad')
    return out


def wgrad_bf16(x, dy, *, k, dil=1, slope=1.0, x_affine=None, out=None):
    """Conv1d weight gradient dwf [k][C_in][C_out] from bf16 operands with fp32 accumulation (v2w_wgrad_bf16): x and dy both fp32 (rounded
    on the way in) or both bf16 tensors.  Returns None when the layer shape has no bf16 instantiation (the caller runs `wgrad`)."""
    B, ci, Lq = x.shape
    co = dy.shape[1]
    if x.dtype != dy.dtype or x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('v2w_wgrad_bf16: x and dy must both be float32 or both bfloat16')
    lib = _hip.load()
    ns = lib.v2w_wgrad_bf16_slabs(B, ci, co, Lq, k)
    if ns == 0:
        return None
    if out is None:
        out = torch.empty((k, ci, co), device=x.device, dtype=torch.float32)
    slab = torch.empty((ns * k * ci * co,), device=x.device, dtype=torch.float32)
    xa, xs = (x_affine[0].data_ptr(), x_affine[1].data_ptr()) if x_affine is not None else (None, None)
    rc = lib.v2w_wgrad_bf16(x.data_ptr(), xa, xs, dy.data_ptr(), out.data_ptr(), slab.data_ptr(), B, ci, co, Lq, k, dil, slope,
                            3 if x.dtype == torch.bfloat16 else 0, _stream(x))
    if rc == _hip.E_SHAPE:
        return None
    _hip.check(rc, 'v2w_wgrad_bf16')
    return out


def convt1d_dgrad(dy, wf, out, *, k, u, mask=None, mask_slope=1.0, algo=ALGO_AUTO):
    """Input gradient of the fused lrelu -> ConvTranspose1d(k, stride u, pad (k-u)/2): dx = lrelu'(x) * sum over the u
    output phases of a small Conv1d on that phase of dy.  dy (B, C_out, u*L), wf [k][C_in][C_out], out (B, C_in, L)."""
    pad = (k - u) // 2
    L = out.shape[2]
    B, co = dy.shape[0], dy.shape[1]
    # float4-aligned lengths: de-interleave dy once into its u phases stacked along the channels (B, u*C_out, L) - one streaming pass -
    # so that every phase conv reads unit-stride rows through the vector staging path (the strided form stages element by element and
    # ran at less than half the forward's rate)
    dyp = None
    if algo == ALGO_AUTO and L % 4 == 0 and conv_tile_config(B, co, out.shape[1], L, 3) is not None:
        dyp = torch.empty((B, u * co, L), device=dy.device, dtype=torch.float32)
        _hip.check(_hip.load().v2w_phase_split(dy.data_ptr(), dyp.data_ptr(), B, co, co, L * u, 1, u, 0, 0, _stream(dy)), 'v2w_phase_split')
    for r in range(u):
        t0, c = (r + pad) % u, (r + pad) // u
        nt = (k - t0 + u - 1) // u
        wr = gather_transpose(wf, t0, u, nt)                   # [nt][C_out][C_in]
        if dyp is not None:
            conv1d(dyp[:, r * co:(r + 1) * co, :], wr, None, out, k=nt, dil=1, slope=1.0, accumulate=(r > 0), wp=pack_mfma(wr), mask=mask,
                   mask_slope=mask_slope, pad_left=c, in_ct=u * co, algo=algo)
        else:
            conv1d(dy, wr, None, out, k=nt, dil=1, slope=1.0, accumulate=(r > 0), wp=pack_mfma(wr), mask=mask, mask_slope=mask_slope,
                   in_stride=u, in_phase=r, pad_left=c, L=L, algo=algo)
    return out


# ---------------------------------------------------------------------------------------------------------------
# backward building blocks
def cbn_backward(dx, xr, gb, stats, running_mean, running_var, *, training, eps=1e-5, sync=None):
    """CondBN backward: returns (dxr, dgb).  `sync(csum)` all-reduces the per-channel sums in data-parallel runs."""
    B, Cc, L = dx.shape
    dev = dx.device
    lib = _hip.load()
    s12 = torch.empty((2 * B * Cc,), device=dev)
    dgb = torch.empty((B, 2 * Cc), device=dev)
    csum = torch.empty((2 * Cc,), device=dev, dtype=torch.float64)
    st = _stream(dx)
    _hip.check(lib.v2w_cbn_bwd_sums(dx.data_ptr(), xr.data_ptr(), gb.data_ptr(), _hip.ptr(stats), running_mean.data_ptr(),
                                    running_var.data_ptr(), s12.data_ptr(), dgb.data_ptr(), csum.data_ptr(), B, Cc, L,
                                    int(training), eps, st), 'v2w_cbn_bwd_sums')
    if sync is not None and training:
        sync(csum)
    tab = torch.empty((B * Cc + 2 * Cc,), device=dev)
    dxr = torch.empty_like(dx)
    _hip.check(lib.v2w_cbn_bwd_apply(dx.data_ptr(), xr.data_ptr(), gb.data_ptr(), _hip.ptr(stats), csum.data_ptr(),
                                     running_mean.data_ptr(), running_var.data_ptr(), tab.data_ptr(), dxr.data_ptr(), B, Cc, L,
                                     int(training), eps, st), 'v2w_cbn_bwd_apply')
    return dxr, dgb


def tail_backward(dy, y, x, wf, *, k, slope):
    """tanh + conv_post backward: returns (dx, dwf [k][C_in][1], dp) - d bias = dp.sum()."""
    B, ci, L = x.shape
    dev = x.device
    dp = torch.empty((B, 1, L), device=dev)
    part = torch.empty((ci * k * 512,), device=dev, dtype=torch.float64)
    dx = torch.empty_like(x)
    dwf = torch.empty((k, ci, 1), device=dev)
    _hip.check(_hip.load().v2w_tail_bwd(dy.data_ptr(), y.data_ptr(), x.data_ptr(), wf.data_ptr(), dp.data_ptr(), part.data_ptr(),
                                        dx.data_ptr(), dwf.data_ptr(), B, ci, L, k, slope, _stream(x)), 'v2w_tail_bwd')
    return dx, dwf, dp


def wn_backward(dwf, v, g, transposed):
    """(dwf [k][C_in][C_out], weight_v, weight_g | None) -> (dv, dg | None)."""
    k, ci, co = dwf.shape
    dv = torch.empty_like(v)
    dg = torch.empty_like(g) if g is not None else None
    _hip.check(_hip.load().v2w_wn_bwd(dwf.data_ptr(), v.data_ptr(), _hip.ptr(g), dv.data_ptr(), _hip.ptr(dg), ci, co, k,
                                      int(transposed), _stream(dwf)), 'v2w_wn_bwd')
    return dv, dg


def cond_backward(dgb, z, sn_w, sn_u, sn_v, sigma, spk, noise):
    """One stage's conditioning backward -> (d weight_orig, d layer.bias, d fcs.weight, d fcs.bias)."""
    B, R = dgb.shape
    dev = dgb.device
    d_w = torch.empty_like(sn_w); d_b = torch.empty((R,), device=dev)
    D = spk.shape[1] + noise.shape[1]
    d_fw = torch.empty((128, D), device=dev); d_fb = torch.empty((128,), device=dev)
    ws = torch.empty((B * 128 + 1,), device=dev)
    _hip.check(_hip.load().v2w_cond_bwd(dgb.data_ptr(), z.data_ptr(), sn_w.data_ptr(), sn_u.data_ptr(), sn_v.data_ptr(),
                                        sigma.data_ptr(), spk.data_ptr(), noise.data_ptr(), d_w.data_ptr(), d_b.data_ptr(),
                                        d_fw.data_ptr(), d_fb.data_ptr(), ws.data_ptr(), B, R // 2, spk.shape[1], noise.shape[1],
                                        _stream(dgb)), 'v2w_cond_bwd')
    return d_w, d_b, d_fw, d_fb


def channel_sum(x):
    """sum over (B, L) per channel of x (B, C, L) -> (C,) fp32: bias gradients (fp64 accumulation inside)."""
    B, Cc, L = x.shape
    stats = torch.empty((2 * Cc + 1,), device=x.device, dtype=torch.float64)
    part = torch.empty((2 * Cc * _hip.V2W_BN_SPLITS,), device=x.device, dtype=torch.float64)
    bn_stats(x, stats, part)
    return stats[:Cc].float()
