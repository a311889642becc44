"""MPD / MSD discriminators on the MI355X, forward and backward.

Mirror of /root/reference/vec2wav/models.py:158-275 - `DiscriminatorP`, `MultiPeriodDiscriminator(hp)`, `DiscriminatorS`,
`MultiScaleDiscriminator()`, the same `forward(y, y_hat) -> (y_d_rs, y_d_gs, fmap_rs, fmap_gs)`, identical `state_dict` keys and
shapes (weight_norm: bias / weight_g / weight_v with Conv2d (k, 1) shapes for the period discriminators; legacy spectral_norm on
the first scale discriminator: bias / weight_orig / weight_u / weight_v) - so `do_%08d` checkpoints load.

Differentiable: each discriminator call is one `autograd.Function` (`_DiscFn`) whose backward runs on the same C ABI (input
gradients on the forward conv kernel with transposed tap-flipped weights, weight gradients on `v2w_wgrad_slice`, weight-norm
backward on `v2w_wn_bwd`), so both optimisation steps of train.py:188-215 work.  No PyTorch/CPU fallback: the convolutions run on
the f32 MFMA tile kernel through the C ABI as stride-1 problems (csrc/v2w_disc.hip explains the mapping):
  stride-s layers  -> `v2w_phase_split` + a conv over the s stacked phases (ceil(k/s)-ish taps),
  (k, 1) Conv2d    -> Conv1d with dilation = period on the flattened (H * period) axis, feature maps kept as (B, C, H, period),
  grouped Conv1d   -> one problem per group on channel slices (`in_ct` / `out_ct`), four groups per launch,
  C_in = 1 layers  -> `v2w_unfold1` (k shifted rows, padded to 16) + a 1-tap conv,
  leaky_relu       -> the conv epilogue (`out_slope`): feature maps are stored activated, as the reference returns them,
  row pitch        -> feature maps live in (B, C, roundup4(length)) buffers (the kernel's float4 staging) and are returned as
                      `[:, :, :length]` views: same shapes and values as the reference, strided when length % 4 != 0.
Weight preparation (weight-norm fold on the HIP kernel; spectral-norm power iteration, tap re-indexing for the stride / group
forms with torch index ops on the weight tensors) is cached per parameter version.
"""
from __future__ import annotations

import math
import os
from typing import List

import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _hip, hipops
from .synthetic import DISC_P_LAYERS, DISC_P_POST, DISC_S_LAYERS, DISC_S_POST

LRELU_SLOPE = 0.1   # models.py:9
_WGRAD_GROUPED = os.environ.get('V2W_DISC_WGRAD_GROUPS', '1') == '1'   # all groups of a layer in one wgrad launch (grid.z)
_UNFOLD_ROWS = 16   # C_in = 1 layers: the k shifted copies padded to the MFMA kernel's smallest channel block


class _DiscConv(nn.Module):
    """Parameter holder of one (weight- or spectral-)normed conv of a discriminator; `wshape` is the reference's weight shape
    (C_out, C_in / groups, k) or (C_out, C_in, k, 1)."""

    def __init__(self, c_in, c_out, k, stride, groups, padding, spectral, conv2d):
        super().__init__()
        self.c_in, self.c_out, self.k, self.stride, self.groups, self.padding = c_in, c_out, k, stride, groups, padding
        self.spectral = spectral
        wshape = (c_out, c_in // groups, k, 1) if conv2d else (c_out, c_in // groups, k)
        fan_in = c_in // groups * k
        bound = 1.0 / math.sqrt(fan_in)
        w = torch.empty(wshape).uniform_(-bound, bound)
        self.bias = nn.Parameter(torch.empty(c_out).uniform_(-bound, bound))
        if spectral:
            self.weight_orig = nn.Parameter(w)
            self.register_buffer('weight_u', F.normalize(torch.randn(c_out), dim=0, eps=1e-12))
            self.register_buffer('weight_v', F.normalize(torch.randn(fan_in), dim=0, eps=1e-12))
        else:
            self.weight_g = nn.Parameter(w.flatten(1).norm(dim=1).view(c_out, *([1] * (len(wshape) - 1))).clone())
            self.weight_v = nn.Parameter(w)
        self._cache = None
        self._pair_rec = None
        self._last_sn = None
        self._jmap = None
        self._wpad = None
        # 'f32' (exact) or 'f16x3': the forward and input-gradient convs of the layers the split-f16 kernel serves - dense, stride 1, an odd
        # tap count: the 1024 -> 1024 five-tap convs that are 54 % of a period discriminator's and 30 % of a scale discriminator's FLOPs -
        # run as f16 hi + lo operands (three MFMAs per product, fp32 accumulate: ~1e-6 of the fp32 result); weight gradients stay exact.
        # Set through `set_precision(module, ...)`.
        self.precision = 'f32'
        # optional form of the short strided ungrouped convs (DiscriminatorP k = 5, stride 3): taps unfolded into channels of a
        # 1-tap conv (exact MAC count, no halo).  Measured equal to the phase-stacked default (40.6 vs 40.4 ms per MPD forward):
        # one tap per staged chunk makes the kernel staging-bound, which cancels the 6/5 tap-slot saving.
        self.unfolded = c_in > 1 and stride > 1 and groups == 1 and k <= 8 and os.environ.get('V2W_DISC_UNFOLD', '0') == '1'

    def extra_repr(self):
        return f'{self.c_in}, {self.c_out}, k={self.k}, stride={self.stride}, groups={self.groups}, ' \
               f'{"spectral_norm" if self.spectral else "weight_norm"}'

    # -- weights in kernel form --------------------------------------------------------------------------------------
    def _folded(self, out=None):
        """-> wf [k][C_in / groups][C_out], normalisation applied (one power iteration first for spectral_norm in training)."""
        co, cig, k = self.c_out, self.c_in // self.groups, self.k
        if not self.spectral:
            return hipops.fold_conv_weight(self.weight_v.detach().reshape(co, cig, k), self.weight_g.detach().reshape(co, 1, 1), out=out)
        w = self.weight_orig.detach()
        wm = w.reshape(co, -1)
        if self.training:      # legacy torch.nn.utils.spectral_norm: v <- norm(W^T u), u <- norm(W v), in place, per forward
            self.weight_v.copy_(F.normalize(torch.mv(wm.t(), self.weight_u), dim=0, eps=1e-12))
            self.weight_u.copy_(F.normalize(torch.mv(wm, self.weight_v), dim=0, eps=1e-12))
        sigma = torch.dot(self.weight_u, torch.mv(wm, self.weight_v))
        self._last_sn = (sigma, self.weight_u.clone(), self.weight_v.clone())
        return hipops.fold_conv_weight((w / sigma).reshape(co, cig, k), None, out=out)

    def invalidate_weight_cache(self):
        self._cache = None

    def kernel_weights(self):
        """Per-group stride-1 weights: dict(wf=[G] of [k'][s * C_in/G][C_out/G], wp=[G] packed or None, kp, pad_left_taps).
        With j - P = s*q + r:  wf'[q + Q][r * cig + c][o] = wf[s*q + r + P][c][o]  (0 where the tap does not exist)."""
        params = list(self.parameters()) + list(self.buffers())
        key = tuple((p.data_ptr(), p._version) for p in params)
        # train mode refolds every call, as the reference's weight-norm / spectral-norm hooks do (and as the generator does): a
        # `.data` mutation (EMA swap, re-initialisation) bumps neither the pointer nor the version counter, so the cache is an
        # eval-mode optimisation only (`invalidate_weight_cache()` after a `.data` edit in eval mode)
        if self._cache is not None and self._cache[0] == key and not self.training:
            return self._cache[1]
        # inside ONE forward(y, y_hat) of a multi-discriminator the weight-normed layers fold once for both inputs (`_pairwise`): nothing
        # can change a parameter between the two calls, and at the reference's batch_size = 2 the ten-odd small launches of this function
        # per layer and call are most of a discriminator forward's wall clock.  (Spectral norm iterates u, v on EVERY call, as the legacy
        # hook does: never shared.)
        if _PAIR.on and not self.spectral and self._pair_rec is not None and self._pair_rec[0] == key:
            return self._pair_rec[1]
        k, s, P, G = self.k, self.stride, self.padding, self.groups
        cig, cog = self.c_in // G, self.c_out // G
        stacked = not self.unfolded and self.c_in > 1
        dev = self.bias.device
        if stacked and (self._wpad is None or self._wpad.device != dev):
            # constants of the re-indexing, built once: the tap map j[q][r] (k = "no such tap" -> the zero row of wpad)
            self._Q = -(-P // s)
            self._kp = self._Q + (k - 1 - P) // s + 1
            q = torch.arange(self._kp, device=dev).view(self._kp, 1) - self._Q
            r = torch.arange(s, device=dev).view(1, s)
            j = s * q + r + P
            self._jmap = torch.where((j >= 0) & (j < k), j, torch.full_like(j, k)).reshape(-1)
            self._wpad = torch.zeros((k + 1, cig, self.c_out), device=dev)
        wf = self._folded(self._wpad[:k] if stacked else None)            # [k][cig][co]
        if self.unfolded:                                    # rows (j, c): the taps become channels of a 1-tap conv
            groups, kp, Q = [wf.reshape(1, k * cig, self.c_out)], 1, 0
        elif self.c_in == 1:                                 # unfolded: rows = taps
            w2 = torch.zeros((1, _UNFOLD_ROWS, self.c_out), device=dev)
            w2[0, :k] = wf[:, 0, :]
            groups, kp, Q = [w2], 1, 0
        else:
            kp, Q, wpad = self._kp, self._Q, self._wpad
            w5 = wpad[self._jmap].reshape(kp, s * cig, G, cog)             # rows (r, c), columns (g, o)
            groups = w5.permute(2, 0, 1, 3).contiguous()                   # [G][kp][s * cig][cog]
        if not torch.is_tensor(groups):
            groups = torch.stack(groups, 0)
        packable = groups.shape[2] % 16 == 0 and (groups.shape[3] % 32 == 0 or groups.shape[3] == 16)
        rec = dict(w4=groups, wf=list(groups.unbind(0)), kp=kp, Q=Q,
                   wp=list(hipops.pack_mfma_batch(groups).unbind(0)) if packable else [None] * groups.shape[0])
        if self.split_eligible(kp, groups.shape[2], groups.shape[3]):
            # (hi, lo) f16 fragments + scale record of the forward conv, per group.  The kernel pipelines over an odd tap count: the
            # two-tap phase-stacked layers (k = 5, stride 3) get a zero third tap behind the others (same pad_left; 1.5 x the MFMAs at 2.2 x the rate)
            ws = groups if kp % 2 == 1 else torch.cat([groups, torch.zeros_like(groups[:, :1])], 1)
            rec['ws4'], rec['kps'] = ws, ws.shape[1]
            rec['wps'] = [hipops.pack_split(ws[g]) for g in range(ws.shape[0])]
        self._cache = (key, rec)
        if _PAIR.on and not self.spectral:
            self._pair_rec = (key, rec)
        return rec

    def split_eligible(self, kp, cigp, cog):
        """The split-f16 kernel serves this layer's stride-1 form: `cigp` stacked input channels and `cog` output channels per group."""
        return (self.precision == 'f16x3' and not self.unfolded and self.c_in > 1 and kp >= 2
                and hipops.split_supported(cigp, cog))

    def param_grads(self, db, dws, sn):
        """Gradients of this layer's parameters (parameters() order) from the bias gradient and the per-group weight gradients
        of the stacked stride-1 form, dws[g] [k'][s * C_in/G][C_out/G] (the inverse of `kernel_weights`' re-indexing)."""
        k, s, G = self.k, self.stride, self.groups
        co, cig = self.c_out, self.c_in // G
        dev = db.device
        if self.c_in == 1:
            dwf = dws[0][0, :k, :].reshape(k, 1, co)
        elif self.unfolded:
            dwf = dws[0].reshape(k, cig, co)
        else:
            kp = dws[0].shape[0]
            d5 = torch.stack(dws, 2).reshape(kp * s, cig, co)            # rows (q, r), then c; columns (g, o)
            dwf = torch.zeros((k + 1, cig, co), device=dev).index_add_(0, self._jmap, d5)[:k]
        dwf = dwf.contiguous()
        if not self.spectral:
            v, g = self.weight_v.detach(), self.weight_g.detach()
            dv, dg = hipops.wn_backward(dwf, v.reshape(co, cig, k), g.reshape(co, 1, 1), False)
            return [db, dg.reshape(g.shape), dv.reshape(v.shape)]
        # W = weight_orig / sigma, sigma = u^T W v with u, v constants (the power iteration runs under no_grad)
        sigma, u, vv = sn
        w = self.weight_orig.detach()
        dW = dwf.permute(2, 1, 0).reshape(w.shape)
        dot = (dW * w).sum() / (sigma * sigma)
        return [db, dW / sigma - dot * torch.outer(u, vv).reshape(w.shape)]


def _check_cuda(*xs):
    for x in xs:
        if not x.is_cuda:
            raise RuntimeError('the discriminators run on the MI355X HIP path only (no CPU fallback)')


def _pitch(n):
    """Row pitch of a feature-map buffer: the MFMA kernel's float4 staging wants rows that are multiples of 4 floats."""
    return (n + 3) // 4 * 4


def _stream(x):
    return torch.cuda.current_stream(x.device).cuda_stream


def _stacked_input(layer: _DiscConv, x, L_in, inner):
    """The stride-1 form's input of a layer: x (B, C_in, pitch) activated feature-map buffer (valid [:L_in * inner]) ->
    (xs (B, C_in', P), U): phases stacked (stride > 1), taps unfolded (optional) or x itself with its tail zeroed (stride 1)."""
    B, pin = x.shape[0], x.shape[2]
    s, G = layer.stride, layer.groups
    lib, st = _hip.load(), _stream(x)
    if s > 1 and layer.unfolded:
        U = (L_in + 2 * layer.padding - layer.k) // s + 1
        P = _pitch(U * inner)
        xs = torch.empty((B, layer.k * layer.c_in, P), device=x.device)
        _hip.check(lib.v2w_unfold_taps(x.data_ptr(), xs.data_ptr(), B, layer.c_in, L_in, inner, s, layer.k, layer.padding, pin, P, st),
                   'v2w_unfold_taps')
    elif s > 1:
        U = -(-L_in // s)
        P = _pitch(U * inner)
        xs = torch.empty((B, s * layer.c_in, P), device=x.device)
        _hip.check(lib.v2w_phase_split(x.data_ptr(), xs.data_ptr(), B, layer.c_in, layer.c_in // G, L_in, inner, s, pin, P, st),
                   'v2w_phase_split')
    else:
        U, xs = L_in, x
        _hip.check(lib.v2w_zero_tail(x.data_ptr(), B * layer.c_in, pin, L_in * inner, st), 'v2w_zero_tail')
    return xs, U


def _conv_layer(layer: _DiscConv, rec, x, L_in, inner, out_slope, keep=None):
    """x (B, C_in, pitch(L_in * inner)) -> (out buffer (B, C_out, pitch(U * inner)), U), U = L_out of the strided conv.  The convs
    run at L = pitch: the tail columns are ordinary positions to the kernel, hold zeros on the input side (phase_split / unfold
    fill them; `v2w_zero_tail` before a stride-1 layer reads a conv output directly) and are never part of the returned views."""
    B = x.shape[0]
    G = layer.groups
    xs, U = _stacked_input(layer, x, L_in, inner)
    if keep is not None:
        keep.append(xs)
    out = torch.empty((B, layer.c_out, xs.shape[2]), device=x.device)
    cig, cog = xs.shape[1] // G, layer.c_out // G
    kw = dict(k=rec['kp'], dil=1 if rec['kp'] == 1 else inner, slope=1.0, pad_left=rec['Q'] * inner, out_slope=out_slope)
    bias = layer.bias.detach()

    def problems(split):
        if split:       # precision = 'f16x3': the split-f16 kernel on the (zero-padded to an odd tap count) stacked weights
            kws = dict(kw, k=rec['kps'], algo=hipops.ALGO_SPLIT)
            return [(xs, rec['ws4'][g], bias[g * cog:(g + 1) * cog], out,
                     dict(kws, wps=rec['wps'][g], group=(g, cig, cog) if G > 1 else None)) for g in range(G)]
        return [(xs, rec['wf'][g], bias[g * cog:(g + 1) * cog], out, dict(kw, wp=rec['wp'][g], group=(g, cig, cog) if G > 1 else None))
                for g in range(G)]

    def run(probs):
        if G == 1:
            x0, w0, b0, o0, k0 = probs[0]
            hipops.conv1d(x0, w0, b0, o0, **k0)
        else:
            for i in range(0, G, 4):
                hipops.conv1d_multi(probs[i:i + 4])

    if 'wps' in rec:
        try:
            run(problems(True))
            return out, U
        except _hip.HipLibraryError as e:       # a shape the split kernel declines (a halo beyond its staging slots): exact from here on
            if e.code != _hip.E_SHAPE:
                raise
            for key in ('wps', 'ws4', 'kps'):
                rec.pop(key, None)
    run(problems(False))
    return out, U


def _unfold_first(layer: _DiscConv, x, T, H, inner):
    B = x.shape[0]
    U = (H + 2 * layer.padding - layer.k) // layer.stride + 1
    if U < 1:
        raise RuntimeError('discriminator input is shorter than the first kernel')
    P = _pitch(U * inner)
    xu = torch.empty((B, _UNFOLD_ROWS, P), device=x.device)
    _hip.check(_hip.load().v2w_unfold1(x.data_ptr(), xu.data_ptr(), B, T, H, inner, layer.stride, layer.k, layer.padding, _UNFOLD_ROWS,
                                       P, _stream(x)), 'v2w_unfold1')
    return xu, U


def _first_layer(layer: _DiscConv, rec, x, T, H, inner):
    """C_in = 1: x (B, 1, T) -> activated buffer (B, C_out, pitch(U * inner)) through the unfolded 1-tap form."""
    xu, U = _unfold_first(layer, x, T, H, inner)
    out = torch.empty((x.shape[0], layer.c_out, xu.shape[2]), device=x.device)
    hipops.conv1d(xu, rec['wf'][0], layer.bias.detach(), out, k=1, dil=1, slope=1.0, wp=rec['wp'][0], out_slope=LRELU_SLOPE)
    return out, U


class _DiscBase(nn.Module):
    """Shared driver of DiscriminatorP / DiscriminatorS: `inner` columns per row (the period, or 1)."""

    def _layers(self):
        return list(self.convs) + [self.conv_post]

    def _geometry(self, t):
        raise NotImplementedError

    def _run(self, x, keep=False):
        """x (B, 1, T) -> state for the views / the backward: per layer (buffer (B, C, pitch), rows U) + the weights used."""
        b, c, t = x.shape
        inner, H = self._geometry(t)
        layers = self._layers()
        recs = [l.kernel_weights() for l in layers]             # (spectral norm: the power iteration of this call happens here)
        sn = [l._last_sn for l in layers]
        bufs = []
        xss = [None] if keep else None                           # stacked inputs kept for the weight gradients (288 GB of HBM)
        f, U = _first_layer(layers[0], recs[0], x, t, H, inner)
        bufs.append((f, U))
        for i in range(1, len(layers)):
            f, U = _conv_layer(layers[i], recs[i], f, U, inner, LRELU_SLOPE if i + 1 < len(layers) else 0.0, xss)
            bufs.append((f, U))
        return dict(bufs=bufs, recs=recs, sn=sn, inner=inner, H=H, T=t, xss=xss)

    def _views(self, st, b):
        inner = st['inner']
        if inner == 1:
            return [f[:, :, :U] for f, U in st['bufs']]
        return [f[:, :, :U * inner].view(b, f.shape[1], U, inner) for f, U in st['bufs']]      # strided when U * inner % 4 != 0

    @_hip.on_tensor_device
    def forward(self, x):
        _check_cuda(x)
        params = [p for l in self._layers() for p in l.parameters()]
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            fmap = list(_DiscFn.apply(self, x, *params))
        else:
            with torch.no_grad():
                x = x.detach().contiguous().float()
                fmap = self._views(self._run(x), x.shape[0])
        return torch.flatten(fmap[-1], 1, -1), fmap


class _DiscFn(torch.autograd.Function):
    """One discriminator call under autograd: forward = `_DiscBase._run`, backward below (the D step and the G step of
    train.py:188-215 both differentiate through it).  Inputs: x and every parameter in `_layers()` order."""

    @staticmethod
    def forward(ctx, disc, x, *params):
        xd = x.detach().contiguous().float()
        st = disc._run(xd, keep=False)        # (the phase-stacked layer inputs are rebuilt in the backward from the maps below them: -11 GB)
        views = tuple(disc._views(st, xd.shape[0]))
        # the maps go through save_for_backward, not a ctx attribute: autograd then releases them when the backward has run (unless
        # retain_graph) - as a plain attribute they lived as long as ANY tensor downstream of this call (train.py keeps `loss_disc_*` and the
        # score lists until the next iteration assigns them: the D step's 19.5 GB of maps stayed allocated through the whole G step)
        ctx.save_for_backward(xd, *[f for f, _u in st['bufs']])
        ctx.rows = [u for _f, u in st['bufs']]
        st['bufs'] = None
        ctx.disc, ctx.st = disc, st
        ctx.need_dx = x.requires_grad
        ctx.need_dw = any(p.requires_grad for p in params)
        return views

    @staticmethod
    @_hip.on_tensor_device
    def backward(ctx, *gouts):
        disc = ctx.disc
        x, *maps = ctx.saved_tensors
        st = dict(ctx.st, bufs=list(zip(maps, ctx.rows)))
        del maps
        lib, stream = _hip.load(), _stream(x)
        layers = disc._layers()
        inner, H, T = st['inner'], st['H'], st['T']
        B, dev = x.shape[0], x.device
        n = len(layers)
        grads = [None] * n           # per layer: tuple of parameter gradients in parameters() order
        dnext, dx = None, None       # gradient wrt the activated map of layer l arriving from layer l + 1: a pitched buffer, or
        merge = None                 # the phase-stacked input gradient of a strided layer + its geometry (merged inside disc_dz)
        for l in reversed(range(n)):
            layer, rec = layers[l], st['recs'][l]
            f, U = st['bufs'][l]
            P, valid, C = f.shape[2], st['bufs'][l][1] * inner, layer.c_out
            g = gouts[l]
            if g is None and dnext is None and merge is None:
                continue                                     # nothing flows through this layer (nor, so far, below it)
            if g is not None:
                g = g.contiguous().float()
            dz = torch.empty((B, C, P), device=dev)
            slope = LRELU_SLOPE if l + 1 < n else 1.0
            rowsum = torch.empty((B * C,), device=dev) if ctx.need_dw else None     # bias gradient partials from the same pass
            if merge is not None:
                dxs_up, cg_up, s_up = merge
                _hip.check(lib.v2w_disc_dz_merge(f.data_ptr(), _hip.ptr(g), dxs_up.data_ptr(), dz.data_ptr(), _hip.ptr(rowsum), B, C, cg_up, U,
                                                 inner, s_up, dxs_up.shape[2], P, slope, stream), 'v2w_disc_dz_merge')
            else:
                _hip.check(lib.v2w_disc_dz(f.data_ptr(), _hip.ptr(g), _hip.ptr(dnext), dz.data_ptr(), _hip.ptr(rowsum), B * C, P, valid, slope,
                                           stream), 'v2w_disc_dz')
            dnext, merge = None, None
            db = None
            if ctx.need_dw:
                db = torch.empty((C,), device=dev)
                _hip.check(lib.v2w_rowsum_reduce(rowsum.data_ptr(), db.data_ptr(), B, C, stream), 'v2w_rowsum_reduce')
            # the layer's stride-1 input (the phases of the map below stacked along the channels; the 16-row unfold of the first layer): rebuilt
            # here, one streaming pass, and only when a weight gradient reads it - the forward keeps the feature maps alone
            G = layer.groups
            if l == 0:
                xs_c = _UNFOLD_ROWS
            else:
                xs_c = layer.c_in * (layer.stride if layer.stride > 1 and not layer.unfolded else (layer.k if layer.stride > 1 else 1))
            xs = None
            if ctx.need_dw:
                if st['xss'] is not None and l > 0:
                    xs = st['xss'][l]
                elif l == 0:
                    xs = _unfold_first(layer, x, T, H, inner)[0]
                else:
                    xs = _stacked_input(layer, st['bufs'][l - 1][0], st['bufs'][l - 1][1], inner)[0]
                assert xs.shape[1] == xs_c and xs.shape[2] == P
            cigp, cog = xs_c // G, C // G
            kp, Q = rec['kp'], rec['Q']
            dil = 1 if kp == 1 else inner
            # ---- weight gradient in the stacked form, then back to the reference's (C_out, C_in / groups, k)
            if ctx.need_dw:                   # (frozen discriminators - `frozen()` around the G step - skip all of this)
                if C == 1:
                    dwp = torch.empty((kp, xs_c, 1), device=dev)
                    _hip.check(lib.v2w_cout1_wgrad(xs.data_ptr(), dz.data_ptr(), dwp.data_ptr(), B, xs_c, P, kp, dil, Q, stream),
                               'v2w_cout1_wgrad')
                    dws = [dwp]
                else:
                    ns = lib.v2w_wgrad_slabs(B, cigp, cog, P)
                    if ns == 0:
                        raise _hip.HipLibraryError(f'v2w_wgrad_slice: no configuration for C_in={cigp}, C_out={cog}')
                    dwg = torch.empty((G, kp, cigp, cog), device=dev)
                    if _WGRAD_GROUPED:
                        slab = torch.empty((G * lib.v2w_wgrad_group_slabs(B, cigp, cog, P, G) * kp * cigp * cog,), device=dev)
                        _hip.check(lib.v2w_wgrad_groups(xs.data_ptr(), dz.data_ptr(), dwg.data_ptr(), slab.data_ptr(), B, cigp, cog, P, kp, dil,
                                                        Q, G, stream), 'v2w_wgrad_groups')
                    else:
                        slab = torch.empty((ns * kp * cigp * cog,), device=dev)
                        for gi in range(G):
                            _hip.check(lib.v2w_wgrad_slice(xs.data_ptr() + gi * cigp * P * 4, dz.data_ptr() + gi * cog * P * 4,
                                                           dwg[gi].data_ptr(), slab.data_ptr(), B, cigp, cog, P, kp, dil, Q, xs_c, C,
                                                           stream), 'v2w_wgrad_slice')
                    dws = list(dwg.unbind(0))
                grads[l] = layer.param_grads(db, dws, st['sn'][l])
            # ---- input gradient: the forward conv kernel with the transposed, tap-flipped weights
            if l > 0 or ctx.need_dx:
                dxs = torch.empty((B, xs_c, P), device=dev)
                if 'wT' not in rec:                                            # shared by the y / y_hat calls of one step
                    packable = cog % 16 == 0 and (cigp % 32 == 0 or cigp == 16)
                    wT4 = rec['w4'].flip(1).transpose(2, 3).contiguous()        # [G][kp][cog][cigp], taps reversed
                    rec['wT'] = list(wT4.unbind(0))
                    rec['wTp'] = list(hipops.pack_mfma_batch(wT4).unbind(0)) if packable else [None] * G
                if 'wps' in rec and 'wTs' not in rec and hipops.split_supported(cog, cigp) and getattr(layer, '_no_split_dgrad', None) != (cog, cigp, P):
                    # the same layer's input-gradient conv in split-f16 form: transposed, tap-reversed (the zero pad tap comes first)
                    wTs4 = rec['ws4'].flip(1).transpose(2, 3).contiguous()       # [G][kps][cog][cigp]
                    rec['wTs4'], rec['wTs'] = wTs4, [hipops.pack_split(wTs4[gi]) for gi in range(G)]
                if 'wTs' in rec:
                    kps = rec['kps']
                    probs = [(dz, rec['wTs4'][gi], None, dxs, dict(k=kps, dil=dil, slope=1.0, pad_left=(kps - 1 - Q) * dil, algo=hipops.ALGO_SPLIT,
                                                                   wps=rec['wTs'][gi], group=(gi, cog, cigp) if G > 1 else None)) for gi in range(G)]
                    try:
                        if G == 1:
                            hipops.conv1d(dz, probs[0][1], None, dxs, **probs[0][4])
                        else:
                            for i in range(0, G, 4):
                                hipops.conv1d_multi(probs[i:i + 4])
                    except _hip.HipLibraryError as e:
                        # the split-f16 kernel took the forward but declines the TRANSPOSED problem (another halo, another channel count per
                        # group): the exact fp32 input-gradient conv below - for this step and, remembered ON THE LAYER (`rec` is rebuilt whenever
                        # a parameter version moves, i.e. after every optimizer step), for every later one: no re-pack, no retry.  The exact conv
                        # overwrites whatever groups the declined launches wrote.
                        if e.code != _hip.E_SHAPE:
                            raise
                        layer._no_split_dgrad = (cog, cigp, P)
                        for key in ('wTs', 'wTs4'):
                            rec.pop(key, None)
                if 'wTs' not in rec:
                    probs = [(dz, rec['wT'][gi], None, dxs, dict(k=kp, dil=dil, slope=1.0, pad_left=(kp - 1 - Q) * dil, wp=rec['wTp'][gi],
                                                                 group=(gi, cog, cigp) if G > 1 else None)) for gi in range(G)]
                    for i in range(0, G, 4):
                        hipops.conv1d_multi(probs[i:i + 4])
                if l == 0:
                    dx = torch.empty((B, 1, T), device=dev)
                    _hip.check(lib.v2w_fold1(dxs.data_ptr(), dx.data_ptr(), B, T, H, inner, layer.stride, layer.k, layer.padding,
                                             _UNFOLD_ROWS, P, stream), 'v2w_fold1')
                elif layer.stride > 1 and not layer.unfolded:
                    merge = (dxs, layer.c_in // G, layer.stride)           # un-stacked by the next iteration's disc_dz
                elif layer.stride > 1:
                    raise NotImplementedError('backward of the unfolded-tap form (V2W_DISC_UNFOLD=1) is not built')
                else:
                    dnext = dxs
        flat = []
        for l in range(n):
            npar = len(list(layers[l].parameters()))
            flat.extend(grads[l] if grads[l] is not None else [None] * npar)
        return (None, dx if ctx.need_dx else None, *flat)


class DiscriminatorP(_DiscBase):
    """models.py:158-193."""

    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        if kernel_size != 5 or stride != 3:
            raise NotImplementedError('DiscriminatorP (HIP): the reference configuration kernel_size=5, stride=3')
        self.period = period
        self.convs = nn.ModuleList([_DiscConv(ci, co, k, s, 1, p, use_spectral_norm, True) for ci, co, k, s, p in DISC_P_LAYERS])
        ci, co, k, s, p = DISC_P_POST
        self.conv_post = _DiscConv(ci, co, k, s, 1, p, use_spectral_norm, True)

    def _geometry(self, t):
        p = self.period
        H = -(-t // p)                   # the reflect pad to a multiple of the period (models.py:176-181) happens inside unfold1
        if H * p - t >= t:
            raise RuntimeError('reflect padding needs an input longer than the pad')
        return p, H


class _PairState(threading.local):
    """`on`: this THREAD is inside a multi-discriminator's (y, y_hat) pairs (see _DiscConv.kernel_weights).  Thread-local and counted: replicas
    driven from several threads (DataParallel) or a nested call do not switch the sharing off under each other."""
    depth = 0

    @property
    def on(self):
        return self.depth > 0


_PAIR = _PairState()


# (y, y_hat) of a weight-normed discriminator as ONE call on the batch [y; y_hat] (the stacks hold no batch statistics: every sample's
# result is what the two calls give; the spectral-normed discriminator iterates u, v once per call and keeps its two calls).  Half the
# launches of a discriminator step's forwards and backwards, and no gradient accumulation over the two calls: train.py's own batch_size = 2 is
# launch-bound.  Only when the parameters ask for gradients - with frozen parameters the real half of two calls needs no backward at all.
# 'auto': up to _BATCH_PAIRS_MAX_SAMPLES input samples per half (above, the autograd slices of the split feature maps cost more than the launches).
BATCH_PAIRS = 'auto'
_BATCH_PAIRS_MAX_SAMPLES = 1 << 20


def _batch_pair(d, y, y_hat):
    if BATCH_PAIRS is False or y.shape != y_hat.shape or any(l.spectral for l in d._layers()):
        return False
    if not (torch.is_grad_enabled() and any(p.requires_grad for p in d.parameters())):
        return False
    return BATCH_PAIRS is True or y.numel() <= _BATCH_PAIRS_MAX_SAMPLES


def _pairwise(discs, inputs):
    """Run every discriminator on its (y, y_hat) pair -> (scores_real, scores_generated, fmaps_real, fmaps_generated)."""
    _PAIR.depth += 1
    try:
        outs = []
        for d, (y, y_hat) in zip(discs, inputs):
            if _batch_pair(d, y, y_hat):
                nb = y.shape[0]
                score, fmap = d(torch.cat([y, y_hat], 0))
                outs.append(((score[:nb], [f[:nb] for f in fmap]), (score[nb:], [f[nb:] for f in fmap])))
            else:
                outs.append((d(y), d(y_hat)))
    finally:
        _PAIR.depth -= 1
        for d in discs:
            for l in d._layers():
                l._pair_rec = None
    return ([r[0] for r, _ in outs], [g[0] for _, g in outs], [r[1] for r, _ in outs], [g[1] for _, g in outs])


class MultiPeriodDiscriminator(nn.Module):
    """models.py:196-216: one DiscriminatorP per period of `hp.periods`, each applied to y then y_hat."""

    def __init__(self, hp):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorP(prd) for prd in hp.periods])

    def forward(self, y, y_hat):
        return _pairwise(self.discriminators, [(y, y_hat)] * len(self.discriminators))


class DiscriminatorS(_DiscBase):
    """models.py:219-243."""

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        self.convs = nn.ModuleList([_DiscConv(ci, co, k, s, g, p, use_spectral_norm, False) for ci, co, k, s, g, p in DISC_S_LAYERS])
        ci, co, k, s, g, p = DISC_S_POST
        self.conv_post = _DiscConv(ci, co, k, s, g, p, use_spectral_norm, False)

    def _geometry(self, t):
        return 1, t


def _avg_pool_fwd(x):
    B, C, L = x.shape
    out = torch.empty((B, C, L // 2 + 1), device=x.device)
    _hip.check(_hip.load().v2w_avgpool4(x.data_ptr(), out.data_ptr(), B * C, L, _stream(x)), 'v2w_avgpool4')
    return out


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return _avg_pool_fwd(x.detach().contiguous().float())

    @staticmethod
    @_hip.on_tensor_device
    def backward(ctx, g):
        B, C, L = ctx.shape
        g = g.contiguous().float()
        dx = torch.empty((B, C, L), device=g.device)
        _hip.check(_hip.load().v2w_avgpool4_bwd(g.data_ptr(), dx.data_ptr(), B * C, L, _stream(g)), 'v2w_avgpool4_bwd')
        return dx


@_hip.on_tensor_device
def avg_pool(x):
    """AvgPool1d(4, 2, padding=2) of models.py:255-258 on (B, 1, L); differentiable."""
    _check_cuda(x)
    if torch.is_grad_enabled() and x.requires_grad:
        return _AvgPoolFn.apply(x)
    with torch.no_grad():
        return _avg_pool_fwd(x.detach().contiguous().float())


class _MeanPool(nn.Module):
    def forward(self, x):
        return avg_pool(x)


class MultiScaleDiscriminator(nn.Module):
    """models.py:246-275: a spectral-normed and two weight-normed DiscriminatorS on the 1x, 1/2x, 1/4x mean-pooled audio."""

    def __init__(self):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorS(use_spectral_norm=True), DiscriminatorS(), DiscriminatorS()])
        self.meanpools = nn.ModuleList([_MeanPool(), _MeanPool()])

    def forward(self, y, y_hat):
        pyramid = [(y, y_hat)]
        for pool in self.meanpools:
            pyramid.append((pool(pyramid[-1][0]), pool(pyramid[-1][1])))
        return _pairwise(self.discriminators, pyramid)


class frozen:
    """Context manager: the parameters of the given discriminators do not require grad inside.  train.py's generator step
    (train.py:201-215) back-propagates through MPD / MSD only to reach `y_g_hat`; the discriminator parameter gradients it also
    produces are discarded by the next `optim_d.zero_grad()`.  Wrapping that step's discriminator forwards in
    `with frozen(mpd, msd):` gives the same training trajectory while the real-audio branch needs no backward at all and the
    generated branch only its input gradients."""

    def __init__(self, *modules):
        self.params = [p for m in modules for p in m.parameters()]

    def __enter__(self):
        self.state = [p.requires_grad for p in self.params]
        for p in self.params:
            p.requires_grad_(False)
        return self

    def __exit__(self, *exc):
        for p, r in zip(self.params, self.state):
            p.requires_grad_(r)
        return False


class _L1MeanFn(torch.autograd.Function):
    """mean |real - fake| of one feature-map pair.  torch's own graph for `torch.mean(torch.abs(real - fake))` keeps the difference and its
    absolute value alive until the backward - two more tensors per feature map, 18 GB over the 2 x 48 maps of a generator step at
    B = 32 x 81 920 samples; here nothing but the two maps (alive anyway: they are what the discriminators return) is saved, the sign is
    rebuilt in the backward."""

    @staticmethod
    def forward(ctx, real, fake):
        ctx.save_for_backward(real, fake)
        return (real - fake).abs_().mean()

    @staticmethod
    def backward(ctx, g):
        real, fake = ctx.saved_tensors
        d = torch.sign(real - fake).mul_(g / real.numel())
        return (d if ctx.needs_input_grad[0] else None), (d.neg() if ctx.needs_input_grad[1] else None)


def set_precision(module, precision):
    """precision of the discriminator convs under `module` (a DiscriminatorP / DiscriminatorS / MultiPeriodDiscriminator /
    MultiScaleDiscriminator): 'f32' (exact, default) or 'f16x3' (see _DiscConv.precision)."""
    if precision not in ('f32', 'f16x3'):
        raise ValueError(f"discriminator precision must be 'f32' or 'f16x3', got {precision!r}")
    for m in module.modules():
        if isinstance(m, _DiscConv):
            m.precision = precision
            m.invalidate_weight_cache()
    return module


def feature_loss(fmap_r, fmap_g):
    """2 x the sum over every feature map of mean |real - generated| (models.py:278-284)."""
    terms = [_L1MeanFn.apply(real, fake) for maps_r, maps_g in zip(fmap_r, fmap_g) for real, fake in zip(maps_r, maps_g)]
    return 2 * sum(terms)


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    """LSGAN discriminator loss: sum over discriminators of mean (1 - D(y))^2 + mean D(y_hat)^2; also the per-discriminator
    values as Python floats (models.py:287-299)."""
    real_terms = [torch.mean((1 - score) ** 2) for score in disc_real_outputs]
    fake_terms = [torch.mean(score ** 2) for score in disc_generated_outputs]
    total = sum(r + f for r, f in zip(real_terms, fake_terms))
    return total, [t.item() for t in real_terms], [t.item() for t in fake_terms]


def generator_loss(disc_outputs):
    """LSGAN generator loss: sum over discriminators of mean (1 - D(y_hat))^2, and the terms (models.py:302-310)."""
    terms = [torch.mean((1 - score) ** 2) for score in disc_outputs]
    return sum(terms), terms
