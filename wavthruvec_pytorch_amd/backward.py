"""Backward pass of the Vec2Wav generator on the HIP path (`loss_gen_all.backward()` of
vec2wav/train.py:214 back-propagates through `generator(wv_feat, spk_emb, noise)`).

`GeneratorFunction` wraps `Generator._forward_hip(save=...)` for autograd.  The backward mirrors the forward schedule in
reverse and runs entirely through the C ABI:

  conv / transposed-conv input gradients  the forward tile kernel itself (v2w_conv1d_fwd) on the output gradient with
                                          transposed(-flipped) weights, the leaky_relu derivative as an epilogue mask
  weight gradients                        v2w_wgrad (MFMA, reduction over positions, deterministic slab reduce)
  bias gradients                          v2w_bn_stats (per-channel sums)
  Conditional BatchNorm                   v2w_cbn_bwd_sums / v2w_cbn_bwd_apply (+ one all-reduce of 2C sums when data-parallel)
  tanh + conv_post                        v2w_tail_bwd
  weight norm, spectral-norm Linear, fcs  v2w_wn_bwd, v2w_cond_bwd

Scope: ResBlock2 (the reference default) and ResBlock1 generators with up to 3 residual branches per stage.
"""
from __future__ import annotations

import torch

from . import _hip, hipops

LRELU_SLOPE = 0.1


class GeneratorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gen, names, x, spk, nz, *params):
        if gen.num_kernels > 3:
            raise NotImplementedError('Generator (HIP) backward supports up to 3 residual branches per stage')
        save = {}
        y = gen._forward_hip(x, spk, nz, save)
        ctx.gen, ctx.names, ctx.saved = gen, names, save
        ctx.needs = [p.requires_grad for p in params]
        ctx.need_dx = x.requires_grad           # the latent as an autograd citizen (the reference's modules are: models.py:116-123)
        return y

    @staticmethod
    @_hip.on_tensor_device
    def backward(ctx, dy):
        grads = generator_backward(ctx.gen, ctx.saved, dy.contiguous().float(), need_dx=ctx.need_dx)
        ctx.saved = None
        out = [grads.get(n) if need else None for n, need in zip(ctx.names, ctx.needs)]
        return (None, None, grads.get('__x__'), None, None, *out)


def _dconv(gen, x, wT, out, *, k, **kw):
    """Input-gradient convolution with the transposed(-flipped) weights wT [k][C_out][C_in]: the forward conv kernel of the
    generator's precision mode (exact fp32 MFMA, or the split-f16 / bf16 kernel when the layer shape has one)."""
    co = wT.shape[2]
    if gen.precision != 'f32' and (k & 1) and k >= 3 and hipops.split_supported(wT.shape[1], co) and co >= gen.split_min_channels \
            and kw.get('in_stride', 0) <= 1:
        bf = gen.precision == 'bf16'
        return hipops.conv1d(x, None, None, out, k=k, algo=hipops.ALGO_BF16 if bf else hipops.ALGO_SPLIT,
                             wps=hipops.pack_split(wT, bf16=bf), **kw)
    return hipops.conv1d(x, wT, None, out, k=k, wp=hipops.pack_mfma(wT), **kw)


def _wgrad(gen, x, dy, *, k, dil, slope, x_affine=None):
    """Conv1d weight gradient in the generator's arithmetic: bf16 operands with fp32 accumulation when the generator computes in bf16 (what
    the reference's autocast backward does, train.py:167,214) and the layer shape has that kernel, the exact fp32 kernel otherwise."""
    if gen.precision == 'bf16':
        dwf = hipops.wgrad_bf16(x, dy, k=k, dil=dil, slope=slope, x_affine=x_affine)
        if dwf is not None:
            return dwf
    return hipops.wgrad(x, dy, k=k, dil=dil, slope=slope, x_affine=x_affine)


def _wn_grads(grads, name, m, dwf):
    """dW in the [k][C_in][C_out] layout -> gradients of the layer's weight_v / weight_g (or plain weight)."""
    if m.weight_normed:
        dv, dg = hipops.wn_backward(dwf, m.weight_v.detach(), m.weight_g.detach(), m.transposed)
        grads[name + '.weight_v'], grads[name + '.weight_g'] = dv, dg
    else:
        dv, _ = hipops.wn_backward(dwf, m.weight.detach(), None, m.transposed)
        grads[name + '.weight'] = dv


@torch.no_grad()
def generator_backward(gen, sv, dy, need_dx=False):
    """dy (B, 1, L_out) -> {parameter name: gradient}.  `sv` is the dict filled by `Generator._forward_hip(save=...)`.
    need_dx: also the gradient w.r.t. the latent input x (key '__x__'): conv_pre's input-gradient conv, one more launch."""
    ws, wf = sv['ws'], sv['wf']
    # the folded weights live in module-owned buffers that the NEXT forward's fold overwrites: harmless while the parameters are unchanged
    # (the same values again); parameters that changed in between - an optimizer step between this graph's forward and its backward -
    # are what autograd itself refuses ("modified by an inplace operation")
    # (`gen`: the fold generation - every fold, a replayed launch plan's included, bumps it.  Unchanged: the buffers are this forward's.  Changed:
    # they were folded again, from whatever the parameters held then - the same values unless a parameter's version moved since the forward)
    if sv.get('vers') is not None and gen._fold_key.get('gen') != sv.get('gen') and gen._param_versions() != sv['vers']:
        raise RuntimeError('Generator (HIP) backward: the generator\'s weights were modified and re-folded by a later forward before this '
                           'backward ran (the saved forward used the earlier weights)')
    x, spk, nz, y, training = sv['x'], sv['spk'], sv['nz'], sv['y'], sv['training']
    B = sv['B']
    dev = x.device
    ns, nk = gen.num_upsamples, gen.num_kernels
    grads = {}

    # ---- tanh + conv_post (models.py:143-145)
    xs_last = ws[f'act.rb{ns - 1}']
    dxs, dwf_post, dp = hipops.tail_backward(dy, y, xs_last, wf['conv_post'], k=7, slope=0.01)
    grads['conv_post.bias'] = hipops.channel_sum(dp)
    _wn_grads(grads, 'conv_post', gen.conv_post, dwf_post)

    z_all = ws['z_ws'].view(ns, B, 128)
    side = gen._side_stream(dev)
    for i in reversed(range(ns)):
        up = gen.ups[i]
        C = up.out_channels
        xr = ws[f'act.up{i}']
        aff = (ws[f'bn.a{i}'], ws[f'bn.s{i}'])
        cur_in = ws['act.pre'] if i == 0 else ws[f'act.rb{i - 1}']
        Lo = xr.shape[2]

        # ---- mean over the nk residual branches: every branch receives dr = dxs / nk
        inv = torch.full((B, C), 1.0 / nk, device=dev)
        zero = torch.zeros((B, C), device=dev)
        from .models import ResBlock1, ResBlock2
        # the merged launches run on ALGO_MFMA, which has no direct-kernel fallback: every gradient conv of every branch must have a
        # tile configuration at ITS kernel size and dilation (a wide halo, e.g. k = 11 with dilation 7, has none: per-branch path)
        merged = gen.precision in ('f32', 'bf16') and gen.algo == hipops.ALGO_AUTO and 1 < nk <= 3 \
            and all(isinstance(gen.resblocks[i * nk + j], ResBlock2) for j in range(nk)) \
            and all(hipops.conv_tile_config(B * nk, C, C, Lo, gen.resblocks[i * nk + j].kernel_size, c.dilation) is not None
                    for j in range(nk) for c in gen.resblocks[i * nk + j].convs)
        if merged:
            # dr is never materialised on this path: the gradient convs read dxs through the per-(b, c) affine (1/nk, 0) - operand and
            # residual - and the quantities that are linear in dr (conv2's weight and bias gradients) are scaled afterwards
            dr = dxs
            db2 = None          # (the per-channel sum of dxs: a memory-bound pass nothing downstream waits for - side stream, below)
        else:
            dr = hipops.affine_apply(dxs, inv, zero, torch.empty_like(dxs))
            db2 = hipops.channel_sum(dr)
        dx = torch.empty_like(dxs)
        if merged:
            # ---- ResBlock2, the forward's launch structure mirrored: the three branches' conv2 input gradients in ONE launch, the conv1
            # input gradients of branches 0 .. nk-2 in one launch and the last branch adding them (heaviest kernel size first), so the
            # tile shape is chosen for three problems' worth of tiles instead of one
            rbs = [gen.resblocks[i * nk + j] for j in range(nk)]
            names = [f'resblocks.{i * nk + j}' for j in range(nk)]
            order = sorted(range(nk), key=lambda j: -rbs[j].kernel_size)
            t1s = [ws[f'act.t1_{i}_{j}'] for j in range(nk)]
            dt1s = [torch.empty_like(dxs) for _ in range(nk)]
            # fragment streams of the gradient convs straight from the forward-layout weights (no transposed copies: C -> C layers)
            # (built by the forward's batched weight fold, beside the forward streams, when the layer is one of its C -> C residual convs)
            wpd = sv.get('wpd', {})
            # the generator's bf16 arithmetic (precision = 'bf16': the reference under torch.autocast): the wide stages' gradient convs on the bf16
            # kernel the forward used - transposed, tap-reversed fragments packed here - the narrow stages' on the exact fp32 tile kernel
            use_bf = gen.precision == 'bf16' and ((C >= gen.split_min_channels and hipops.split_supported(C, C)) or (C == 32 and Lo % 4 == 0)) \
                and all(rb.kernel_size >= 3 and (rb.kernel_size & 1) for rb in rbs)

            def wsel(nm):
                """Kernel and weights of the input-gradient conv of layer `nm`."""
                if use_bf:
                    return dict(algo=hipops.ALGO_BF16, wps=hipops.pack_split(hipops.transpose_flip(wf[nm]), bf16=True))
                return dict(algo=hipops.ALGO_MFMA, wp=wpd.get(nm) if wpd.get(nm) is not None else hipops.pack_mfma_dgrad(wf[nm]))
            w2s, w1s = [wsel(names[j] + '.convs.1') for j in range(nk)], [wsel(names[j] + '.convs.0') for j in range(nk)]
            p2, p1 = [q.get('wp') for q in w2s], [q.get('wp') for q in w1s]
            # r_j = t1 + conv2(lrelu(t1)) + b2   ->   dt1 = dr + lrelu'(t1) * conv(dr; W2^T flipped)
            # ... and, from the same launch's epilogue, the per-tile channel sums of dt1_j = the bias gradient of conv1_j
            # (the launch's tile shape follows its WIDEST halo - the wide-halo tile variants are other shapes: probe with that branch)
            # narrow stages (C = 32 / 16): both input-gradient convs of all branches in ONE kernel (v2w_stage_args::bwd_*) - dxs read once,
            # every dt1_j written once and not read back, the branch sum in registers: 9 tensor passes instead of 18
            fused, ntile, rsp = False, 0, None
            if C in gen.fuse_stage and C in (16, 32) and gen.fuse_stage_backward and all(q is not None for q in p1 + p2):
                ks_, dd2, dd1 = [rb.kernel_size for rb in rbs], [rb.convs[1].dilation for rb in rbs], [rb.convs[0].dilation for rb in rbs]
                ntile = hipops.resblock2_stage_bwd_rows(B, C, Lo, ks_, dd2, dd1)
                if ntile:
                    rsp = [torch.empty((ntile * C * 2,), device=dev) for _ in range(nk)]       # (tile, wave) channel sums of dt1_j: conv1_j's bias gradient
                    fused = hipops.resblock2_stage(
                        dxs, (inv, zero), [dict(wp1=p2[j], b1=None, wp2=p1[j], b2=None, k=ks_[j], dil1=dd2[j], dil2=dd1[j]) for j in range(nk)],
                        dx, slope=1.0, out_div=0.0, bwd=(t1s, dt1s, xr, aff, LRELU_SLOPE, rsp))
            if not fused:
                kw, dw = max(((rb.kernel_size, rb.convs[1].dilation) for rb in rbs), key=lambda kd: kd[1] * (kd[0] - 1))
                ntile = hipops.conv_rowsum_tiles(B, nk, C, C, Lo, kw, dw) if Lo % 4 == 0 and not use_bf else 0      # (an epilogue of the f32 tile kernel)
                rsp = [torch.empty((ntile * C * 2,), device=dev) if ntile else None for _ in range(nk)]
            if not fused:
                hipops.conv1d_multi([(dxs, None, None, dt1s[j],
                                      dict(k=rbs[j].kernel_size, dil=rbs[j].convs[1].dilation, slope=1.0, in_affine=(inv, zero), res=dxs,
                                           res_affine=(inv, zero), mask=(t1s[j], None), mask_slope=LRELU_SLOPE, rowsum=rsp[j], **w2s[j]))
                                     for j in order])
            # t1 = x + conv1(lrelu(x)) + b1, x = a*xr + s   ->   dx = sum_j dt1_j + lrelu'(x) * conv(dt1_j; W1^T flipped)
            def dconv1(j, out, **extra):
                return (dt1s[j], None, None, out,
                        dict(k=rbs[j].kernel_size, dil=rbs[j].convs[0].dilation, slope=1.0, res=dt1s[j], mask=(xr, aff), mask_slope=LRELU_SLOPE,
                             **w1s[j], **extra))
            if not fused:
                parts = [torch.empty_like(dxs) for _ in range(nk - 1)]
                hipops.conv1d_multi([dconv1(j, parts[j]) for j in order if j < nk - 1])
                hipops.conv1d_multi([dconv1(nk - 1, dx, add=parts)])
            # weight / bias gradients: nothing downstream waits for them - side stream, beside the next stage's gradient convs
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            def branch_grads(j, db2):
                c1, c2 = rbs[j].convs[0], rbs[j].convs[1]
                k = rbs[j].kernel_size
                _wn_grads(grads, names[j] + '.convs.1', c2, _wgrad(gen, t1s[j], dxs, k=k, dil=c2.dilation, slope=LRELU_SLOPE).mul_(1.0 / nk))
                grads[names[j] + '.convs.1.bias'] = db2
                _wn_grads(grads, names[j] + '.convs.0', c1, _wgrad(gen, xr, dt1s[j], k=k, dil=c1.dilation, slope=LRELU_SLOPE, x_affine=aff))
                if ntile:
                    st = torch.empty((2 * C + 1,), device=dev, dtype=torch.float64)
                    hipops.bn_reduce_partials(rsp[j], ntile, C, B * Lo, st)
                    grads[names[j] + '.convs.0.bias'] = st[:C].float()
                else:
                    grads[names[j] + '.convs.0.bias'] = hipops.channel_sum(dt1s[j])

            # The LAST stage of the walk (stage 0: the widest convs) leaves the side stream a backlog the main stream has nothing left to
            # run beside (it waited ~2.5 ms for it at the end of the backward): its heaviest branch's weight gradients go to the main stream
            on_main = [order[0]] if i == 0 and nk > 1 else []
            with torch.cuda.stream(side):
                made = []
                db2 = hipops.channel_sum(dxs) * (1.0 / nk)
                made.append(db2)
                for j in range(nk):
                    if j in on_main:
                        continue
                    g0 = dict(grads)
                    branch_grads(j, db2)
                    made += [v for kk, v in grads.items() if kk not in g0]
            for t in [dxs, xr, aff[0], aff[1]] + dt1s + t1s + [r_ for r_ in rsp if r_ is not None]:
                t.record_stream(side)
            for t in made:
                t.record_stream(main)
            for j in on_main:       # (db2 is only handed on as the bias gradient here: no kernel of the main stream reads it)
                branch_grads(j, db2)
        for j in range(nk if not merged else 0):
            rb = gen.resblocks[i * nk + j]
            name = f'resblocks.{i * nk + j}'
            k = rb.kernel_size
            if isinstance(rb, ResBlock1):
                # x_{n+1} = x_n + conv2_n(lrelu(u_n)) + b2,  u_n = conv1_n(lrelu(x_n)) + b1,  x_0 = a*xr + s   (models.py:37-44)
                xin = [xr, ws[f'act.xa_{i}_{j}'], ws[f'act.xb_{i}_{j}']]
                dcur = dr
                for n in (2, 1, 0):
                    c1, c2 = rb.convs1[n], rb.convs2[n]
                    u = ws[f'act.t1_{i}_{j}_{n}']
                    x_aff = aff if n == 0 else None
                    w2T = hipops.transpose_flip(wf[f'{name}.convs2.{n}'])
                    du = torch.empty_like(dr)
                    _dconv(gen, dcur, w2T, du, k=k, dil=1, slope=1.0,
                                  mask=(u, None), mask_slope=LRELU_SLOPE)
                    _wn_grads(grads, f'{name}.convs2.{n}', c2, _wgrad(gen, u, dcur, k=k, dil=1, slope=LRELU_SLOPE))
                    grads[f'{name}.convs2.{n}.bias'] = hipops.channel_sum(dcur) if n < 2 else db2
                    w1T = hipops.transpose_flip(wf[f'{name}.convs1.{n}'])
                    if n > 0:
                        dprev = torch.empty_like(dr)
                        _dconv(gen, du, w1T, dprev, k=k, dil=c1.dilation, slope=1.0, res=dcur,
                                      mask=(xin[n], None), mask_slope=LRELU_SLOPE)
                    else:   # into the stage's dx, summed over the branches
                        _dconv(gen, du, w1T, dx, k=k, dil=c1.dilation, slope=1.0, res=dcur,
                                      mask=(xr, aff), mask_slope=LRELU_SLOPE, accumulate=(j > 0))
                        dprev = None
                    _wn_grads(grads, f'{name}.convs1.{n}', c1,
                              _wgrad(gen, xin[n], du, k=k, dil=c1.dilation, slope=LRELU_SLOPE, x_affine=x_aff))
                    grads[f'{name}.convs1.{n}.bias'] = hipops.channel_sum(du)
                    dcur = dprev
                continue
            c1, c2 = rb.convs[0], rb.convs[1]
            t1 = ws[f'act.t1_{i}_{j}']
            # r_j = t1 + conv2(lrelu(t1)) + b2   ->   dt1 = dr + lrelu'(t1) * conv(dr; W2^T flipped)
            w2T = hipops.transpose_flip(wf[name + '.convs.1'])
            dt1 = torch.empty_like(dr)
            _dconv(gen, dr, w2T, dt1, k=k, dil=c2.dilation, slope=1.0, res=dr,
                          mask=(t1, None), mask_slope=LRELU_SLOPE)
            _wn_grads(grads, name + '.convs.1', c2, _wgrad(gen, t1, dr, k=k, dil=c2.dilation, slope=LRELU_SLOPE))
            grads[name + '.convs.1.bias'] = db2
            # t1 = x + conv1(lrelu(x)) + b1, x = a*xr + s   ->   dx += dt1 + lrelu'(x) * conv(dt1; W1^T flipped)
            w1T = hipops.transpose_flip(wf[name + '.convs.0'])
            _dconv(gen, dt1, w1T, dx, k=k, dil=c1.dilation, slope=1.0, res=dt1,
                          mask=(xr, aff), mask_slope=LRELU_SLOPE, accumulate=(j > 0))
            _wn_grads(grads, name + '.convs.0', c1,
                      _wgrad(gen, xr, dt1, k=k, dil=c1.dilation, slope=LRELU_SLOPE, x_affine=aff))
            grads[name + '.convs.0.bias'] = hipops.channel_sum(dt1)

        # ---- Conditional BatchNorm (modules.py:20-30): through the affine, the batch statistics and into gamma / beta
        cbn = gen.cbns[i]
        bn = cbn.batch_nrom
        dxr, dgb = hipops.cbn_backward(dx, xr, ws[f'gb.{i}'], ws.get(f'bn.stats{i}'), bn.running_mean, bn.running_var,
                                       training=training, eps=bn.eps, sync=gen.stat_sync)
        ly, fc = cbn.layer, gen.fcs[i]
        sn_u, sn_v = sv['sn_uv'][i]               # the vectors the forward of THIS graph used (ly.weight_u/_v may have moved on)
        # the conditioning branch (spectral-norm Linear + fcs[i]: five small latency-bound kernels per stage) hangs off dgb only and
        # nothing downstream waits for it: it runs on the side stream, beside the convolutions, and is joined at the end
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            d_w, d_b, d_fw, d_fb = hipops.cond_backward(dgb, z_all[i].contiguous(), ly.weight_orig.detach(), sn_u, sn_v,
                                                        ws['sigma_ws'][i:i + 1], spk, nz)
            d_ub = hipops.channel_sum(dxr)       # the upsampler's bias gradient: a memory-bound pass beside the gradient convs
        dgb.record_stream(side)
        dxr.record_stream(side)
        for t in (d_w, d_b, d_fw, d_fb, d_ub):
            t.record_stream(main)
        grads[f'cbns.{i}.layer.weight_orig'], grads[f'cbns.{i}.layer.bias'] = d_w, d_b
        grads[f'fcs.{i}.weight'], grads[f'fcs.{i}.bias'] = d_fw, d_fb

        # ---- leaky_relu -> ConvTranspose1d (models.py:128-129)
        grads[f'ups.{i}.bias'] = d_ub
        _wn_grads(grads, f'ups.{i}', up,
                  hipops.wgrad(cur_in, dxr, k=up.kernel_size, u=up.stride, slope=LRELU_SLOPE))
        dxs = torch.empty_like(cur_in)
        hipops.convt1d_dgrad(dxr, wf[f'ups.{i}'], dxs, k=up.kernel_size, u=up.stride, mask=(cur_in, None), mask_slope=LRELU_SLOPE)

    # ---- conv_pre (models.py:123): no activation in front of it, no input gradient requested
    grads['conv_pre.bias'] = hipops.channel_sum(dxs)
    _wn_grads(grads, 'conv_pre', gen.conv_pre, hipops.wgrad(x, dxs, k=7, dil=1, slope=1.0))
    if need_dx:     # dL/dx = conv(dxs; W_pre^T, taps reversed) - no activation in front of conv_pre, no mask
        dxin = torch.empty_like(x)
        _dconv(gen, dxs, hipops.transpose_flip(wf['conv_pre']), dxin, k=7, dil=1, slope=1.0)
        grads['__x__'] = dxin
    torch.cuda.current_stream(dev).wait_stream(side)
    return grads


class ResBlockFunction(torch.autograd.Function):
    """A stand-alone `ResBlock1` / `ResBlock2` as an autograd citizen (the reference's are: models.py:37-44, 65-70): the forward of
    models._resblock_forward with every step's input kept, the backward from the same entry points the generator's backward uses -
    input-gradient convs with the transposed, tap-reversed weights and the leaky_relu derivative as an epilogue mask, v2w_wgrad, per-channel
    sums, v2w_wn_bwd.  `pairs`: [(conv_a, conv_b | None)] as in _resblock_forward; params: the parameters of every conv, in `names` order."""

    @staticmethod
    def forward(ctx, rb, pairs, names, x, *params):
        from .models import _fold_one
        cur = x.detach().contiguous().float()
        steps = []
        for ca, cb in pairs:
            wfa, wpa = _fold_one(ca, cur.device)
            if cb is None:        # ResBlock2: out = cur + conv_a(lrelu(cur))
                out = torch.empty_like(cur)
                hipops.conv1d(cur, wfa, ca.bias.detach(), out, k=ca.kernel_size, dil=ca.dilation, slope=LRELU_SLOPE, res=cur, wp=wpa)
                steps.append((ca, None, cur, None, wfa, None))
            else:                 # ResBlock1: out = cur + conv_b(lrelu(conv_a(lrelu(cur))))
                u = torch.empty_like(cur)
                hipops.conv1d(cur, wfa, ca.bias.detach(), u, k=ca.kernel_size, dil=ca.dilation, slope=LRELU_SLOPE, wp=wpa)
                wfb, wpb = _fold_one(cb, cur.device)
                out = torch.empty_like(cur)
                hipops.conv1d(u, wfb, cb.bias.detach(), out, k=cb.kernel_size, dil=cb.dilation, slope=LRELU_SLOPE, res=cur, wp=wpb)
                steps.append((ca, cb, cur, u, wfa, wfb))
            cur = out
        ctx.steps, ctx.names, ctx.prefix = steps, names, {id(m): n for n, m in rb.named_modules()}
        ctx.needs = [p.requires_grad for p in params]
        ctx.need_dx = x.requires_grad
        return cur

    @staticmethod
    @_hip.on_tensor_device
    @torch.no_grad()
    def backward(ctx, dout):
        dcur = dout.contiguous().float()
        grads = {}

        def dconv(src, wf, out, m, **kw):
            wT = hipops.transpose_flip(wf)
            return hipops.conv1d(src, wT, None, out, k=m.kernel_size, dil=m.dilation, slope=1.0, wp=hipops.pack_mfma(wT), **kw)

        for ca, cb, cur, u, wfa, wfb in reversed(ctx.steps):
            na = ctx.prefix[id(ca)]
            if cb is None:
                # out = cur + conv_a(lrelu cur) + b  ->  dcur' = dout + lrelu'(cur) * conv(dout; Wa^T flipped)
                _wn_grads(grads, na, ca, hipops.wgrad(cur, dcur, k=ca.kernel_size, dil=ca.dilation, slope=LRELU_SLOPE))
                grads[na + '.bias'] = hipops.channel_sum(dcur)
                dprev = torch.empty_like(dcur)
                dconv(dcur, wfa, dprev, ca, res=dcur, mask=(cur, None), mask_slope=LRELU_SLOPE)
            else:
                nb = ctx.prefix[id(cb)]
                _wn_grads(grads, nb, cb, hipops.wgrad(u, dcur, k=cb.kernel_size, dil=cb.dilation, slope=LRELU_SLOPE))
                grads[nb + '.bias'] = hipops.channel_sum(dcur)
                du = torch.empty_like(dcur)
                dconv(dcur, wfb, du, cb, mask=(u, None), mask_slope=LRELU_SLOPE)
                _wn_grads(grads, na, ca, hipops.wgrad(cur, du, k=ca.kernel_size, dil=ca.dilation, slope=LRELU_SLOPE))
                grads[na + '.bias'] = hipops.channel_sum(du)
                dprev = torch.empty_like(dcur)
                dconv(du, wfa, dprev, ca, res=dcur, mask=(cur, None), mask_slope=LRELU_SLOPE)
            dcur = dprev
        ctx.steps = None
        out = [grads.get(n) if need else None for n, need in zip(ctx.names, ctx.needs)]
        return (None, None, None, dcur if ctx.need_dx else None, *out)
