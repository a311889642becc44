"""ORACLE - test infrastructure only.  Not shipped, not imported by the product path.

A CPU restatement of the reference Vec2Wav ``Generator.forward``
(/root/reference/vec2wav/models.py:116-147, vec2wav/modules.py:20-30) written as a flat
function over a ``state_dict``: every weight-norm fold, spectral-norm step, BatchNorm formula
and affine is spelled out with stock ``torch`` CPU ops (``F.conv1d`` / ``F.conv_transpose1d``),
none of ``torch.nn.utils.weight_norm`` / ``spectral_norm`` / ``nn.BatchNorm1d`` is used.

Who may use this file: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg - as the checker / the reported CPU baseline, never as the thing shipped.
``wavthruvec_pytorch_amd`` never imports it and raises when its HIP library is missing.

Parity pin: the reference has no tests and no golden vectors, so this
restatement is pinned by importing the reference itself in the build container
(``tools/gen_goldens.py``) and committing its outputs as fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this file against every one of them on CPU.

Reference anchors, op by op:
  weight-norm fold  w = g * v / ||v||_(all dims but 0)   torch weight_norm(dim=0); models.py:18-33,58-61,83,90,100
  conv_pre          Conv1d(k=7, pad=3)                    models.py:83,123
  stage i           leaky_relu(0.1) -> ConvTranspose1d(k,u,pad=(k-u)//2)   models.py:128-129, 89-92
  fcs[i]            Linear(spk_dim+noise_dim -> 128)      models.py:109-111,131
  cbns[i]           BatchNorm1d(affine=False) ; legacy spectral_norm(Linear(128->2C)) ; gamma*xhat+beta
                                                          modules.py:14-28
  ResBlock2         x += conv_{k,d}(lrelu(x)), d in dilation[:2]            models.py:65-70
  ResBlock1         x += conv_{k,1}(lrelu(conv_{k,d}(lrelu(x)))), d in (1,3,5)   models.py:37-44
  mean over kernels xs / num_kernels                      models.py:135-141
  tail              leaky_relu(default 0.01) -> conv_post(k=7,pad=3) -> tanh     models.py:143-145
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # models.py:10
BN_EPS = 1e-5      # nn.BatchNorm1d default, modules.py:14
BN_MOMENTUM = 0.1
SN_EPS = 1e-12     # torch.nn.utils.spectral_norm default


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """vec2wav/utils.py:35-36."""
    return int((kernel_size * dilation - dilation) / 2)


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """w = g * v / ||v||, norm over every dim except 0 (per C_out for Conv1d, per C_in for ConvTranspose1d)."""
    norm = v.pow(2).sum(dim=tuple(range(1, v.dim())), keepdim=True).sqrt()
    return v * (g / norm)


def _conv_weight(sd: Dict[str, torch.Tensor], prefix: str, dtype) -> torch.Tensor:
    if prefix + '.weight' in sd:  # after remove_weight_norm (models.py:149-156)
        return sd[prefix + '.weight'].to(dtype)
    return fold_weight_norm(sd[prefix + '.weight_g'].to(dtype), sd[prefix + '.weight_v'].to(dtype))


def spectral_norm_weight(w_orig, u, v, training: bool, eps: float = SN_EPS):
    """Legacy ``torch.nn.utils.spectral_norm`` pre-forward hook, one power iteration in train mode.

    Returns (W / sigma, u_new, v_new); in eval mode u, v are returned unchanged.
    """
    if training:
        with torch.no_grad():     # the hook iterates under no_grad and then treats u, v as constants
            v = F.normalize(torch.mv(w_orig.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(w_orig, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(w_orig, v))
    return w_orig / sigma, u, v


def batch_norm_no_affine(x, running_mean, running_var, training: bool):
    """``nn.BatchNorm1d(C, affine=False)`` on (B,C,L).  Returns (xhat, new_running_mean, new_running_var)."""
    if training:
        n = x.shape[0] * x.shape[2]
        mean = x.mean(dim=(0, 2))
        var_b = x.var(dim=(0, 2), unbiased=False)
        xhat = (x - mean[None, :, None]) * torch.rsqrt(var_b + BN_EPS)[None, :, None]
        var_u = var_b * (n / max(n - 1, 1))
        new_rm = (1 - BN_MOMENTUM) * running_mean + BN_MOMENTUM * mean
        new_rv = (1 - BN_MOMENTUM) * running_var + BN_MOMENTUM * var_u
        return xhat, new_rm, new_rv
    xhat = (x - running_mean[None, :, None]) * torch.rsqrt(running_var + BN_EPS)[None, :, None]
    return xhat, running_mean, running_var


def _resblock(sd, prefix, x, k, dil, resblock1: bool, dtype):
    if resblock1:
        for n in range(3):
            d = dil[n]
            xt = F.leaky_relu(x, LRELU_SLOPE)
            xt = F.conv1d(xt, _conv_weight(sd, f'{prefix}.convs1.{n}', dtype), sd[f'{prefix}.convs1.{n}.bias'].to(dtype),
                          padding=get_padding(k, d), dilation=d)
            xt = F.leaky_relu(xt, LRELU_SLOPE)
            xt = F.conv1d(xt, _conv_weight(sd, f'{prefix}.convs2.{n}', dtype), sd[f'{prefix}.convs2.{n}.bias'].to(dtype),
                          padding=get_padding(k, 1), dilation=1)
            x = xt + x
        return x
    for n in range(2):
        d = dil[n]
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _conv_weight(sd, f'{prefix}.convs.{n}', dtype), sd[f'{prefix}.convs.{n}.bias'].to(dtype),
                      padding=get_padding(k, d), dilation=d)
        x = xt + x
    return x


def generator_forward(sd, h, x, spk_emb, noise, training: bool = True, dtype=torch.float32, probes=None, sync_stats=None):
    """Reference ``Generator.forward`` over a state_dict, without autograd (see ``generator_forward_impl``)."""
    with torch.no_grad():
        return generator_forward_impl(sd, h, x, spk_emb, noise, training, dtype, probes, sync_stats)


def generator_forward_impl(sd: Dict[str, torch.Tensor], h, x, spk_emb, noise, training: bool = True,
                           dtype=torch.float32, probes: Optional[dict] = None, sync_stats=None):
    """Reference ``Generator.forward`` over a state_dict.

    Returns ``(y, new_buffers)``; ``new_buffers`` holds the post-forward values of every buffer
    the reference mutates in train mode (``running_mean/var``, ``num_batches_tracked``,
    ``weight_u/_v``).  ``probes`` (optional dict) receives the named intermediate tensors.
    ``sync_stats(sum, sumsq, count)``, if given, replaces the local batch statistics by
    all-reduced ones (the data-parallel CondBN exchange); test-only hook.
    """
    resblock1 = h.resblock == '1'
    nk = len(h.resblock_kernel_sizes)
    new_buffers = OrderedDict()
    x = x.to(dtype)
    spk_noise = torch.cat((spk_emb, noise), dim=1).to(dtype)

    x = F.conv1d(x, _conv_weight(sd, 'conv_pre', dtype), sd['conv_pre.bias'].to(dtype), padding=3)
    if probes is not None:
        probes['conv_pre'] = x
    for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _conv_weight(sd, f'ups.{i}', dtype), sd[f'ups.{i}.bias'].to(dtype),
                               stride=u, padding=(k - u) // 2)
        if probes is not None:
            probes[f'ups.{i}'] = x
        z = F.linear(spk_noise, sd[f'fcs.{i}.weight'].to(dtype), sd[f'fcs.{i}.bias'].to(dtype))
        # --- ConditionalBatchNorm1d.forward (modules.py:20-30): BN first, then the SN hook fires inside layer().
        p = f'cbns.{i}'
        rm = sd[p + '.batch_nrom.running_mean'].to(dtype)
        rv = sd[p + '.batch_nrom.running_var'].to(dtype)
        if training and sync_stats is not None:
            n_loc = x.shape[0] * x.shape[2]
            s, ss, n = sync_stats(x.double().sum(dim=(0, 2)), x.double().pow(2).sum(dim=(0, 2)), n_loc)
            mean = (s / n)
            var_b = (ss / n - mean * mean).clamp_min(0)
            xhat = ((x.double() - mean[None, :, None]) * torch.rsqrt(var_b + BN_EPS)[None, :, None]).to(dtype)
            new_rm = ((1 - BN_MOMENTUM) * rm.double() + BN_MOMENTUM * mean).to(dtype)
            new_rv = ((1 - BN_MOMENTUM) * rv.double() + BN_MOMENTUM * var_b * (n / max(n - 1, 1))).to(dtype)
        else:
            xhat, new_rm, new_rv = batch_norm_no_affine(x, rm, rv, training)
        w_sn, u_new, v_new = spectral_norm_weight(sd[p + '.layer.weight_orig'].to(dtype),
                                                  sd[p + '.layer.weight_u'].to(dtype),
                                                  sd[p + '.layer.weight_v'].to(dtype), training)
        gb = F.linear(z, w_sn, sd[p + '.layer.bias'].to(dtype))
        gamma, beta = gb.chunk(2, 1)
        x = gamma[:, :, None] * xhat + beta[:, :, None]
        if probes is not None:
            probes[f'cbns.{i}'] = x
        nbt = sd[p + '.batch_nrom.num_batches_tracked']
        new_buffers[p + '.batch_nrom.running_mean'] = new_rm
        new_buffers[p + '.batch_nrom.running_var'] = new_rv
        new_buffers[p + '.batch_nrom.num_batches_tracked'] = nbt + 1 if training else nbt.clone()
        new_buffers[p + '.layer.weight_u'] = u_new
        new_buffers[p + '.layer.weight_v'] = v_new
        # --- multi-receptive-field sum
        xs = None
        for j, (rk, rd) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
            r = _resblock(sd, f'resblocks.{i * nk + j}', x, rk, rd, resblock1, dtype)
            if probes is not None:
                probes[f'resblocks.{i * nk + j}'] = r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)  # default slope 0.01 (models.py:143)
    x = F.conv1d(x, _conv_weight(sd, 'conv_post', dtype), sd['conv_post.bias'].to(dtype), padding=3)
    if probes is not None:
        probes['conv_post'] = x
    return torch.tanh(x), new_buffers


def generator_gradients(sd, h, x, spk_emb, noise, dy, training: bool = True, dtype=torch.float32, want_x: bool = False):
    """Reference gradients of sum(y * dy) w.r.t. every floating-point parameter of the state_dict (torch autograd through
    the restated forward): what `loss_gen_all.backward()` (vec2wav/train.py:214) sends into the generator.  want_x: the gradient
    w.r.t. the latent input as well, under the key '__x__'."""
    leaves = {}
    if want_x:
        x = x.detach().clone().to(dtype).requires_grad_(True)
    for k, v in sd.items():
        if v.is_floating_point() and not any(t in k for t in ('running_', 'weight_u', 'layer.weight_v')):
            leaves[k] = v.detach().clone().to(dtype).requires_grad_(True)
    sd2 = {k: leaves.get(k, v) for k, v in sd.items()}
    y, nb = generator_forward_impl(sd2, h, x, spk_emb, noise, training, dtype)
    (y * dy.to(dtype)).sum().backward()
    grads = {k: v.grad for k, v in leaves.items()}
    if want_x:
        grads['__x__'] = x.grad
    return y.detach(), grads, nb


def apply_buffers(sd: Dict[str, torch.Tensor], new_buffers: Dict[str, torch.Tensor]) -> None:
    """Write the post-forward buffers back into ``sd`` (what the reference does in place)."""
    for k, v in new_buffers.items():
        sd[k] = v.to(sd[k].dtype).clone()


def calibrate_running_stats(sd, h, x, spk_emb, noise, dtype=torch.float32):
    """Calibrated eval: set running stats := batch stats of (x, spk, noise).

    Mirrors the recipe `momentum = 1.0 ; one train forward` used on the reference when the
    goldens were generated: running_mean = batch mean, running_var = unbiased batch var;
    weight_u/_v take one power-iteration step; num_batches_tracked += 1.
    """
    probes = {}
    _, nb = generator_forward(sd, h, x, spk_emb, noise, training=True, dtype=dtype, probes=probes)
    for i in range(len(h.upsample_rates)):
        t = probes[f'ups.{i}']
        n = t.shape[0] * t.shape[2]
        nb[f'cbns.{i}.batch_nrom.running_mean'] = t.mean(dim=(0, 2))
        nb[f'cbns.{i}.batch_nrom.running_var'] = t.var(dim=(0, 2), unbiased=False) * (n / max(n - 1, 1))
    apply_buffers(sd, nb)


def remove_weight_norm_sd(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """State-dict view of ``Generator.remove_weight_norm()``: weight_g/_v -> weight (bias, weight order)."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith('.weight_g'):
            p = k[:-len('.weight_g')]
            out[p + '.weight'] = fold_weight_norm(v, sd[p + '.weight_v'])
        elif k.endswith('.weight_v') and k[:-len('.weight_v')] + '.weight_g' in sd:
            continue
        else:
            out[k] = v
    return out
