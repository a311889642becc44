"""Algorithmic work of one Generator forward (SURVEY.md section 8(d) definitions), used by bench.py's roofline.

FLOPs: conv `2*C_in*C_out*k*L_out*B`; convT `2*C_in*C_out*k*L_in*B`.
Bytes (layer-granular): every conv/convT reads its input activation once and writes its output once, its weights
are read once; activation, bias, residual (same tensor as the operand), CondBN affine and /num_kernels are fused
(0 extra bytes); the 2nd..n-th resblock of a stage additionally reads the running sum once.
"""
from __future__ import annotations

from typing import Dict, List


def conv_layers(h, batch: int, n_frame: int, act_bytes: int = 4) -> List[Dict]:
    """One dict per conv/convT launch of a forward: name, kind, shape, flops, bytes.  act_bytes is the size of an element of
    the activations BETWEEN layers (4: fp32; 2: the bf16 storage mode of precision='bf16'); the latents read by conv_pre, the
    audio written by conv_post and the weights are fp32 in every mode."""
    resblock1 = h.resblock == '1'
    es = 4
    layers = []
    c0 = h.upsample_initial_channel
    L = n_frame

    def act(c, l, e=None):
        return batch * c * l * (act_bytes if e is None else e)

    layers.append(dict(name='conv_pre', kind='conv', cin=h.num_wv_feat, cout=c0, k=7, d=1, L=L,
                       flops=2.0 * h.num_wv_feat * c0 * 7 * L * batch,
                       bytes=act(h.num_wv_feat, L, 4) + act(c0, L) + h.num_wv_feat * c0 * 7 * es))
    nk = len(h.resblock_kernel_sizes)
    cout = c0
    for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
        cin, cout = c0 // 2 ** i, c0 // 2 ** (i + 1)
        layers.append(dict(name=f'ups.{i}', kind='convt', cin=cin, cout=cout, k=k, u=u, L=L,
                           flops=2.0 * cin * cout * k * L * batch,
                           bytes=act(cin, L) + act(cout, L * u) + cin * cout * k * es))
        L = L * u
        for j, (rk, rd) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
            dils = [d for dd in rd[:3] for d in (dd, 1)] if resblock1 else list(rd[:2])
            for n, d in enumerate(dils):
                last = n == len(dils) - 1
                extra = act(cout, L) if (last and j > 0) else 0   # running sum re-read
                layers.append(dict(name=f'resblocks.{i * nk + j}.{n}', kind='conv', cin=cout, cout=cout, k=rk, d=d, L=L,
                                   flops=2.0 * cout * cout * rk * L * batch,
                                   bytes=2 * act(cout, L) + extra + cout * cout * rk * es))
    layers.append(dict(name='conv_post', kind='conv', cin=cout, cout=1, k=7, d=1, L=L,
                       flops=2.0 * cout * 7 * L * batch, bytes=act(cout, L) + act(1, L, 4) + cout * 7 * es))
    return layers


def totals(h, batch: int, n_frame: int, act_bytes: int = 4):
    ls = conv_layers(h, batch, n_frame, act_bytes)
    return sum(l['flops'] for l in ls), sum(l['bytes'] for l in ls)
