"""Vec2Wav `Generator` with the reference's Python surface and a hand-written HIP forward.

Mirrors /root/reference/vec2wav/models.py:13-156 - `Generator(h)`, `forward(x, spk_emb, noise)`,
`remove_weight_norm()`, `ResBlock1`, `ResBlock2`, `LRELU_SLOPE`, identical `state_dict` keys and shapes -
so it drops into `vec2wav/train.py` (`from models import Generator`) and loads `g_%08d` checkpoints.
The modules below only HOLD parameters; `Generator.forward` runs the whole path through the C ABI of
libvec2wav_hip.so (include/vec2wav_hip.h).  There is no PyTorch/CPU fallback: CPU tensors or a missing
library raise.
"""
from __future__ import annotations

import math
import warnings
from typing import Dict, List, Optional

import contextlib
import torch
import torch.nn as nn

from . import _hip, hipops
from .modules import ConditionalBatchNorm1d
from .utils import get_padding, init_weights  # noqa: F401  (re-exported like the reference's models.py)

LRELU_SLOPE = 0.1  # models.py:10
_SIDE_STREAMS: Dict[str, 'torch.cuda.Stream'] = {}
_CAPTURE_STREAMS: Dict[str, 'torch.cuda.Stream'] = {}      # per process and device: the stream HIP graphs are warmed up and captured on
_Z_CHANNEL = 128   # models.py:110


# ------------------------------------------------------------------------------------------------
# parameter holders with the key names torch.nn.utils.weight_norm produces (bias, weight_g, weight_v)
# ------------------------------------------------------------------------------------------------
class _WNConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation, self.padding = kernel_size, stride, dilation, padding
        wshape = (in_channels, out_channels, kernel_size) if self.transposed else (out_channels, in_channels, kernel_size)
        fan_in = wshape[1] * kernel_size
        bound = 1.0 / math.sqrt(fan_in)
        v = torch.empty(wshape).uniform_(-bound, bound)  # nn.Conv default (kaiming_uniform a=sqrt(5))
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        # weight_norm(dim=0): g = ||v|| over all dims but 0, so that w == v at construction
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.weight_v = nn.Parameter(v)

    @property
    def weight_normed(self) -> bool:
        return 'weight_g' in self._parameters

    def remove_weight_norm(self):
        """`torch.nn.utils.remove_weight_norm`: fold g, v into a plain `weight` parameter (keys: bias, weight)."""
        if not self.weight_normed:
            raise ValueError(f'weight_norm of \'weight\' not found in {self}')
        with torch.no_grad():
            v, g = self.weight_v, self.weight_g
            w = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))
        del self._parameters['weight_g']
        del self._parameters['weight_v']
        self.weight = nn.Parameter(w.detach())

    def extra_repr(self):
        s = f'{self.in_channels}, {self.out_channels}, kernel_size=({self.kernel_size},), stride=({self.stride},)'
        if self.padding:
            s += f', padding=({self.padding},)'
        if self.dilation != 1:
            s += f', dilation=({self.dilation},)'
        return s


class Conv1d(_WNConvBase):
    """Holder for `weight_norm(nn.Conv1d(cin, cout, k, 1, dilation=d, padding=get_padding(k, d)))`."""
    transposed = False


class ConvTranspose1d(_WNConvBase):
    """Holder for `weight_norm(nn.ConvTranspose1d(cin, cout, k, u, padding=(k-u)//2))`."""
    transposed = True


class _Shape:
    """Shape carrier for the library's host-only queries (an aligned address that is never dereferenced)."""

    def __init__(self, *shape):
        self.shape = shape

    def data_ptr(self):
        return 4096


def _fold_one(m, device):
    v, g = (m.weight_v.detach(), m.weight_g.detach()) if m.weight_normed else (m.weight.detach(), None)
    wf = hipops.fold_conv_weight(v, g)
    return wf, hipops.pack_mfma(wf)


def _resblock_forward(rb, x, pairs):
    """Standalone residual block (the reference's are callable: models.py:37-44, 65-70): x (B, C, L) fp32 on the GPU.
    pairs: [(conv_a, conv_b | None)]: x = x + conv_b(lrelu(conv_a(lrelu(x))))  or  x = x + conv_a(lrelu(x))."""
    if not x.is_cuda:
        raise RuntimeError('ResBlock (HIP): GPU tensors only; there is no CPU fallback')
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in rb.parameters())):
        raise NotImplementedError('ResBlock (HIP): standalone blocks run without autograd; back-propagate through Generator.forward')
    with torch.no_grad():
        cur = x.detach().contiguous().float()
        for ca, cb in pairs:
            wfa, wpa = _fold_one(ca, cur.device)
            if cb is None:
                out = torch.empty_like(cur)
                hipops.conv1d(cur, wfa, ca.bias.detach(), out, k=ca.kernel_size, dil=ca.dilation, slope=LRELU_SLOPE, res=cur, wp=wpa)
            else:
                t = torch.empty_like(cur)
                hipops.conv1d(cur, wfa, ca.bias.detach(), t, k=ca.kernel_size, dil=ca.dilation, slope=LRELU_SLOPE, wp=wpa)
                wfb, wpb = _fold_one(cb, cur.device)
                out = torch.empty_like(cur)
                hipops.conv1d(t, wfb, cb.bias.detach(), out, k=cb.kernel_size, dil=cb.dilation, slope=LRELU_SLOPE, res=cur, wp=wpb)
            cur = out
        return cur


class ResBlock1(nn.Module):
    """models.py:13-50: three (dilated conv, conv) pairs with residuals."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.h = h
        self.channels, self.kernel_size, self.dilation = channels, kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList([
            Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d))
            for d in dilation[:3]])
        self.convs2 = nn.ModuleList([
            Conv1d(channels, channels, kernel_size, 1, dilation=1, padding=get_padding(kernel_size, 1))
            for _ in range(3)])

    @_hip.on_tensor_device
    def forward(self, x):
        """models.py:37-44 as HIP launches (no-grad; inside `Generator.forward` the block runs merged with its siblings)."""
        return _resblock_forward(self, x, [(c1, c2) for c1, c2 in zip(self.convs1, self.convs2)])

    def remove_weight_norm(self):
        for l in self.convs1:
            l.remove_weight_norm()
        for l in self.convs2:
            l.remove_weight_norm()


class ResBlock2(nn.Module):
    """models.py:53-74: two dilated convs with residuals (uses dilation[0], dilation[1] only)."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.h = h
        self.channels, self.kernel_size, self.dilation = channels, kernel_size, tuple(dilation)
        self.convs = nn.ModuleList([
            Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d))
            for d in dilation[:2]])

    @_hip.on_tensor_device
    def forward(self, x):
        """models.py:65-70 as HIP launches (no-grad; inside `Generator.forward` the block runs merged with its siblings)."""
        return _resblock_forward(self, x, [(c, None) for c in self.convs])

    def remove_weight_norm(self):
        for l in self.convs:
            l.remove_weight_norm()


class Generator(nn.Module):
    """HiFi-GAN-style Vec2Wav generator (reference: vec2wav/models.py:77-156)."""

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        if self.num_upsamples > 8:
            raise ValueError('at most 8 upsample stages are supported')
        c0 = h.upsample_initial_channel
        self.conv_pre = Conv1d(h.num_wv_feat, c0, 7, 1, padding=3)
        resblock = ResBlock1 if h.resblock == '1' else ResBlock2   # models.py:84 (string compare)

        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            self.ups.append(ConvTranspose1d(c0 // (2 ** i), c0 // (2 ** (i + 1)), k, u, padding=(k - u) // 2))

        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for j, (k, d) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
                self.resblocks.append(resblock(h, ch, k, d))

        self.conv_post = Conv1d(ch, 1, 7, 1, padding=3)

        self.cbns = nn.ModuleList()
        self.fcs = nn.ModuleList()
        for i in range(len(self.ups)):
            self.fcs.append(nn.Linear(h.spk_dim + h.noise_dim, _Z_CHANNEL))
            self.cbns.append(ConditionalBatchNorm1d(256 // pow(2, i)))   # hard-coded widths, models.py:113
            if self.cbns[i].num_features != self.ups[i].out_channels:
                raise ValueError('ConditionalBatchNorm1d widths are 256 // 2**i (models.py:113): '
                                 'upsample_initial_channel must be 512')

        # ---- HIP-path state (not part of the state_dict)
        self.algo = hipops.ALGO_AUTO          # hipops.ALGO_DIRECT forces the scalar cross-check kernels
        self.stat_sync = None                 # callable(stats fp64 tensor) -> all-reduced in place (distributed.BNStatSync)
        self.always_refold = True             # train mode: fold weight norm every forward, as the reference's hook does
        self.fuse_pairs = (16,)               # stage widths whose conv pairs run as ONE fused kernel (measured: pays at C=16,
                                              # ties at C=32 where the per-layer tiles are already MFMA-bound)
        self.fuse_stage = (8, 16, 32)         # ResBlock2 stage widths whose WHOLE residual section runs as one kernel
        self.precision = 'f32'                # 'f32': exact fp32 MFMA everywhere (default).  'bf16': bf16 operands, one MFMA per
                                              # product (BASELINE configs[2]).  'f16x3': Conv1d layers with
                                              # C_out >= split_min_channels run on the f16 matrix pipe with split operands
                                              # (x_hi*w_hi + x_hi*w_lo + x_lo*w_hi, fp32 accumulate; hipops.ALGO_SPLIT)
        self.split_min_channels = 64
        self.bf16_storage = True              # precision == 'bf16': keep activations in bf16 between layers (no-grad forwards)
        self.fuse_wide = True                 # bf16 storage: the residual convs of the wide stages (C >= 64) as one launch per conv position
        self.fuse_wide_stage = True           # bf16 storage: the whole residual section of a wide stage (C = 64 / 128 / 256) as ONE kernel
        self.fuse_up = True                   # bf16 storage: the NEXT stage's transposed conv (stride 2 / 4) inside the kernel of a stage (C = 32 .. 256):
                                              # the stage's output never leaves the chip, one launch less per stage
        self.fuse_post = True                 # bf16 storage: leaky_relu -> conv_post -> tanh inside the kernel of the last (C = 16) stage: the stage's
                                              # output (335 MB at configs[2]) is never written nor read back (v2w_stage_bf16_n16.hip, 7-tap tail)
        self.inline_stats = True              # bf16 storage, train mode, no statistics exchange: the BatchNorm sums of a stage's input are added up by the
                                              # PRODUCING kernel (integer atomics) and folded by the CONSUMING stage kernel - no reduce / finalize launches
                                              # between the stage kernels (csrc/v2w_bnacc.h)
        self._split_wide = set()
        self._ws: Dict[str, torch.Tensor] = {}
        self._slabs: Dict[tuple, 'hipops.SplitKSlab'] = {}
        self._wts: Dict[str, torch.Tensor] = {}
        self._fold_key: Dict[str, tuple] = {}
        self._profile = None                  # list -> (tag, start_event, end_event) per conv launch (bench.py roofline)

    # -------------------------------------------------------------------------------------------
    def enable_sync_batchnorm(self, group=None, single_rank_collective=None):
        """Data-parallel CondBN: all-reduce the per-stage batch statistics over `group` (RCCL on GPUs).  A one-rank group exchanges
        nothing unless `single_rank_collective=True` (bench.py --force-pg and the -m gpu tests: the RCCL code path on a one-GPU box)."""
        from .distributed import BNStatSync
        self.stat_sync = BNStatSync(group, single_rank_collective=single_rank_collective)
        return self

    def capture_graph(self, x, spk_emb, noise, warmup: int = 2):
        """Capture one forward (current train/eval mode, these shapes) into a HIP graph and return `run(x, spk_emb, noise)`.

        The forward is ~40-60 kernel launches; at inference sizes (B=1, T~50) it is launch-bound, and replaying one graph
        removes the per-launch host cost.  Inputs are copied into static buffers, the returned tensor is the graph's static
        output (clone it to keep it across calls).  Data-parallel statistics exchange cannot be captured.
        Every buffer the captured launches touch - activations, folded weights, the split-over-C_in scratch (`_slab`) - belongs to
        THIS module, so graphs of different generators may replay concurrently on different streams."""
        if self.stat_sync is not None:
            raise RuntimeError('capture_graph: the RCCL statistics all-reduce cannot be part of a captured graph')
        sx, ss, sn = x.detach().clone().contiguous(), spk_emb.detach().clone().contiguous(), noise.detach().clone().contiguous()
        with torch.no_grad():
            # warm-up and capture run on ONE per-device stream: the module keys its split-over-C_in scratch by stream, and the warm-up
            # must have sized it before the capture (nothing may be allocated while a stream is capturing)
            side = _CAPTURE_STREAMS.get(str(sx.device))
            if side is None:
                side = _CAPTURE_STREAMS[str(sx.device)] = torch.cuda.Stream(device=sx.device)
            side.wait_stream(torch.cuda.current_stream(sx.device))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):     # builds every workspace buffer and the fold plan outside the capture
                    self.forward(sx, ss, sn)
            torch.cuda.current_stream(sx.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                sy = self.forward(sx, ss, sn)

        def run(x, spk_emb, noise):
            if x.dtype == sx.dtype and spk_emb.dtype == ss.dtype and noise.dtype == sn.dtype and x.device == sx.device:
                torch._foreach_copy_([sx, ss, sn], [x, spk_emb, noise])      # the three inputs into the static buffers: ONE launch
            else:
                sx.copy_(x); ss.copy_(spk_emb); sn.copy_(noise)
            graph.replay()
            return sy

        run.graph = graph
        return run

    def remove_weight_norm(self):
        print('Removing weight norm...')
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()

    # -------------------------------------------------------------------------------------------
    @staticmethod
    def _side_stream(device):
        st = _SIDE_STREAMS.get(str(device))      # per process and device, not per module: modules stay deep-copyable / picklable
        if st is None:
            st = _SIDE_STREAMS[str(device)] = torch.cuda.Stream(device=device)
        return st

    def _buf(self, name, shape, dtype=torch.float32, device=None):
        t = self._ws.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device:
            t = torch.empty(shape, device=device, dtype=dtype)
            self._ws[name] = t
        return t

    def _slab(self, device):
        """This module's split-over-C_in scratch for launches on the CURRENT stream of `device` (hipops.SplitKSlab; v2w_conv1d_args::splitk_ws).
        One slab per (module, stream): two generators - or one generator on two streams - never share one, so captured graphs of
        different modules may replay concurrently."""
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        slab = self._slabs.get(key)
        if slab is None:
            slab = self._slabs[key] = hipops.SplitKSlab()
        return slab

    def _wbuf(self, name, shape, dtype=torch.float32, device=None):
        """Folded / packed WEIGHT buffers: like `_buf`, but they stay with the module when a training forward hands its activation
        buffers to the autograd graph - the batched fold's descriptor table (pointers) then survives from step to step instead of being
        rebuilt and copied to the device in front of every training forward.  A backward whose forward's weights have since been
        modified AND re-folded by a later forward is refused (backward.py), as autograd refuses an in-place modified saved tensor."""
        t = self._wts.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device:
            t = torch.empty(shape, device=device, dtype=dtype)
            self._wts[name] = t
        return t

    def _timed(self, tag, fn, *args, **kw):
        """Launch `fn`; when profiling is on, bracket it with events on the launching (current) stream."""
        if self._profile is None:
            return fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*args, **kw)
        e1.record()
        self._profile.append((tag, e0, e1))
        return r

    def _conv_layers(self):
        yield 'conv_pre', self.conv_pre
        for i, m in enumerate(self.ups):
            yield f'ups.{i}', m
        for i, rb in enumerate(self.resblocks):
            if isinstance(rb, ResBlock1):
                for n, m in enumerate(rb.convs1):
                    yield f'resblocks.{i}.convs1.{n}', m
                for n, m in enumerate(rb.convs2):
                    yield f'resblocks.{i}.convs2.{n}', m
            else:
                for n, m in enumerate(rb.convs):
                    yield f'resblocks.{i}.convs.{n}', m
        yield 'conv_post', self.conv_post

    def invalidate_weight_cache(self):
        """Forget the folded / packed weights.  The eval-mode cache follows the parameters' storage pointers and in-place version
        counters; a `.data` mutation (`p.data.copy_(ema)`, `m.weight.data.normal_()`) changes neither - call this after one."""
        self._fold_key.clear()

    def _fold_weights(self, device, need_wf=False, bf16_only=False):
        """K0: weight-norm fold of every conv.  Layers with an MFMA tile configuration are folded AND packed into their
        fragment stream `wp` by one batched call (two launches for the whole generator); the others (conv_post, odd
        shapes, or everything under ALGO_DIRECT) are folded one by one into `wf` [k][C_in][C_out].  Skipped while the
        parameters are unchanged (storage pointers + in-place version counters) unless `always_refold` in train mode."""
        layers = list(self._conv_layers())
        vers = []
        for name, m in layers:
            ps = (m.weight_v, m.weight_g) if m.weight_normed else (m.weight,)
            vers.append(tuple((p.data_ptr(), p._version) for p in ps))
        state = (tuple(vers), self.algo, str(device), need_wf, bf16_only)
        force = (self.training and self.always_refold) or need_wf
        if not force and self._fold_key.get('state') == state:
            return self._fold_key['wf'], self._fold_key['wp']
        wf, wp, batch = {}, {}, []
        wpd = {}        # need_wf: fragment streams of the input-gradient convs of the C -> C residual convs (backward.py), from the same pass
        for name, m in layers:
            if bf16_only and name != 'conv_post':      # bf16 activation storage: every other layer runs on its bf16 fragments (_split_weights):
                wf[name], wp[name] = None, None        # no fp32 fold / fragment stream is read, none is built
                continue
            v, g = (m.weight_v.detach(), m.weight_g.detach()) if m.weight_normed else (m.weight.detach(), None)
            u = m.stride if m.transposed else 1
            mfma_ok = (self.algo != hipops.ALGO_DIRECT and name != 'conv_post' and
                       hipops.conv_tile_config(1, m.in_channels, m.out_channels, 64, m.kernel_size,
                                               1 if m.transposed else m.dilation, u) is not None)
            if mfma_ok and need_wf:     # a forward that will be back-propagated: dgrad / wgrad also read the plain layout - written by the
                # same batched fold (was: three launches per layer, ~100 host-bound launches and 1.6 ms in front of every training forward)
                n_el = m.kernel_size * m.in_channels * m.out_channels
                wfb = self._wbuf('wf.' + name, (m.kernel_size, m.in_channels, m.out_channels), device=device)
                wpb = self._wbuf('wp.' + name, (n_el,), device=device)
                wdb = None
                if not m.transposed and m.in_channels == m.out_channels and name.startswith('resblocks.'):
                    wdb = wpd[name] = self._wbuf('wpd.' + name, (n_el,), device=device)
                batch.append((v, g, wpb, m.in_channels, m.out_channels, m.kernel_size, u, m.transposed, wfb, wdb))
                wf[name], wp[name] = wfb, wpb
            elif mfma_ok:
                wpb = self._wbuf('wp.' + name, (m.kernel_size * m.in_channels * m.out_channels,), device=device)
                batch.append((v, g, wpb, m.in_channels, m.out_channels, m.kernel_size, u, m.transposed))
                wf[name], wp[name] = None, wpb
            else:
                wfb = self._wbuf('wf.' + name, (m.kernel_size, m.in_channels, m.out_channels), device=device)
                scratch = self._wbuf('wf_scratch', (max(2048, m.out_channels, m.in_channels),), device=device)
                (hipops.fold_convt_weight if m.transposed else hipops.fold_conv_weight)(v, g, wfb, scratch)
                wf[name], wp[name] = wfb, None
        if batch:
            key = tuple((q[0].data_ptr(), 0 if q[1] is None else q[1].data_ptr(), q[2].data_ptr(),
                         0 if len(q) < 9 or q[8] is None else q[8].data_ptr()) for q in batch)
            plan = self._fold_key.get('plan')
            if plan is None or plan.key != key:
                plan = hipops.FoldPlan(batch, device)
                plan.key = key
                self._fold_key['plan'] = plan
            plan.run()
        self._fold_key.update(state=state, wf=wf, wp=wp, wpd=wpd, vers=tuple(vers), gen=self._fold_key.get('gen', 0) + 1)
        return wf, wp

    def _split_weights(self, device, all_ups=False, ups_stream=None, events=None, between=None):
        """precision == 'f16x3': (hi, lo) half-precision fragments + scale record of every Conv1d layer the split kernel
        serves.  Follows the fold cache: rebuilt whenever `_fold_weights` rebuilt (train mode: every forward).
        `ups_stream`: the upsamplers' fold + pack launches (ten latency-bound kernels, ~55 us at five stages) go to that stream - the
        caller joins it before the first upsampler; they then run beside the Conv1d batch and conv_pre.
        `events` (a dict) + `between` (a callable): the order on `ups_stream` becomes ups.0 -> between() -> every Conv1d but conv_pre ->
        ups.1 .. ups.n, with an event recorded behind each step ('ups.i', 'rest'): the caller waits for exactly what its next launch
        reads instead of for the whole stream (at B = 32 x T = 256 the stream's 230 us of small kernels outlast conv_pre by 90 us)."""
        if self.precision == 'f32' or self.algo == hipops.ALGO_DIRECT:
            return {}
        if self.precision not in ('f16x3', 'bf16'):
            raise ValueError(f"Generator.precision must be 'f32', 'f16x3' or 'bf16', got {self.precision!r}")
        gen = (self._fold_key.get('gen', 0), self.precision, all_ups)
        cached = self._fold_key.get('wps')
        if cached is not None and cached[0] == gen:
            return cached[1]
        out, batch = {}, []
        self._split_wide = set()          # layers the per-layer split conv kernel takes (the others feed the fused C = 32 / 16 stages)
        picked = []
        for name, m in self._conv_layers():
            if m.transposed or name == 'conv_post':
                continue
            # wide layers: the per-layer split kernel; the C = 32 / 16 ResBlock2 stages: the fused split stage kernel
            stage32 = (m.in_channels == m.out_channels and m.out_channels in (16, 32) and m.out_channels in self.fuse_stage
                       and name.startswith('resblocks.') and ('.convs.' in name or (all_ups and self.precision == 'bf16')))   # (ResBlock1 pairs on bf16 tensors: narrow stages too)
            wide = (m.out_channels >= self.split_min_channels and hipops.split_supported(m.in_channels, m.out_channels)
                    and m.kernel_size % 2 == 1 and m.kernel_size >= 3)       # the split kernel pipelines over an odd tap count
            if wide or stage32:
                picked.append((name, m, wide))
        # ONE arena, slices in execution order: the fused stage kernel walks the six streams of a stage as one contiguous
        # stream; every slice of a wide layer carries the padding unit the conv kernel's stage copies may touch
        sizes = [hipops.split_halves(m.kernel_size, m.in_channels, m.out_channels) if wide
                 else hipops.split_units_halves(m.kernel_size, m.in_channels, m.out_channels) for _n, m, wide in picked]
        arena = self._buf('wps.arena', (sum(sizes) + 1024,), dtype=torch.float16, device=device)
        scs = self._buf('wsc.arena', (4 * max(1, len(picked)),), device=device)
        off = 0
        for i, ((name, m, wide), n) in enumerate(zip(picked, sizes)):
            v, g = (m.weight_v.detach(), m.weight_g.detach()) if m.weight_normed else (m.weight.detach(), None)
            wpsb, scb = arena[off:off + n], scs[4 * i:4 * i + 4]
            off += n
            batch.append((v.contiguous(), g, wpsb, scb))
            out[name] = (wpsb, scb)
            if wide:
                self._split_wide.add(name)
        # two batches: conv_pre alone on the calling stream (the forward needs it at once), every other layer on `ups_stream` beside conv_pre
        first = [q for (nm, _m, _w), q in zip(picked, batch) if nm == 'conv_pre'] if ups_stream is not None else batch
        rest = [q for (nm, _m, _w), q in zip(picked, batch) if nm != 'conv_pre'] if ups_stream is not None else []
        def mark(name):
            if events is not None and ups_stream is not None:
                ev = torch.cuda.Event()
                ev.record(ups_stream)
                events[name] = ev

        def run_plan(slot, sub, strm):
            if not sub:
                return
            key = tuple((v.data_ptr(), 0 if g is None else g.data_ptr(), w.data_ptr()) for (v, g, w, _s) in sub) + (self.precision,)
            plan = self._fold_key.get(slot)
            if plan is None or plan.key != key:
                plan = hipops.SplitPlan(sub, device, bf16=self.precision == 'bf16')
                plan.key = key
                self._fold_key[slot] = plan
            with (torch.cuda.stream(strm) if strm is not None else contextlib.nullcontext()):
                plan.run()

        def fold_up(i, m):
            v, g = (m.weight_v.detach(), m.weight_g.detach()) if m.weight_normed else (m.weight.detach(), None)
            wfb = self._buf(f'wfbf.ups.{i}', (m.kernel_size, m.in_channels, m.out_channels), device=device)
            # (its own scratch: `wf_scratch` is in use by the folds of the main stream)
            scratch = self._buf('wf_scratch_ups', (max(2048, m.out_channels, m.in_channels),), device=device)
            with (torch.cuda.stream(ups_stream) if ups_stream is not None else contextlib.nullcontext()):
                hipops.fold_convt_weight(v, g, wfb, scratch)
                w = hipops.pack_bf16_convt(wfb, m.stride, out=self._ws.get(f'wpsbf.ups.{i}'))
            if w is not None:
                self._ws[f'wpsbf.ups.{i}'] = w
                out[f'ups.{i}'] = w
            mark(f'ups.{i}')

        # the transposed convs run on the bf16 matrix pipe too (v2w_convt1d_bf16_fwd); the narrow upsamplers are memory-side: with fp32
        # tensors the f32 kernel's epilogue moves their bytes faster; with bf16 storage every upsampler runs there
        ups_bf = [(i, m) for i, m in enumerate(self.ups) if self.precision == 'bf16' and (m.out_channels >= 64 or all_ups)]
        run_plan('split_plan', first, None)
        if ups_bf and ups_bf[0][0] == 0:       # what the forward needs first: conv_pre's fragments (above, on its own stream), then ups.0's
            fold_up(*ups_bf.pop(0))
        if between is not None:
            between()
        run_plan('split_plan_rest', rest, ups_stream)
        mark('rest')
        for i, m in ups_bf:
            fold_up(i, m)
        self._fold_key['wps'] = (gen, out)
        return out

    def _bf16_storage_kernels_exist(self, B, T) -> bool:
        """bf16 activation storage needs a bf16-tensor kernel for EVERY layer: asked of the library before the forward starts (shape
        queries only), so that a configuration one of them declines - a residual kernel size > 11 or a halo > 32 positions in a
        narrow stage, a conv the bf16 tile kernel has no configuration for - runs with fp32 tensors between the layers (bf16
        operands, the `bf16_storage = False` arithmetic) instead of failing in the middle of the forward."""
        key = (B, T, self.num_kernels, tuple(self.fuse_stage))
        hit = self._fold_key.get('bf16_storage_ok')
        if hit is not None and hit[0] == key:
            return hit[1]
        ok = hipops.conv_bf16_config(B, 1, self.conv_pre.in_channels, self.conv_pre.out_channels, T, 7, 1, 1, io_bf16=2) is not None
        L, nk = T, self.num_kernels
        for i, up in enumerate(self.ups):
            ok = ok and hipops.conv_bf16_config(B, 1, up.in_channels, up.out_channels, L, up.kernel_size, 1, up.stride, io_bf16=3) is not None
            L *= up.stride
            C = up.out_channels
            rbs = self.resblocks[i * nk:(i + 1) * nk]
            if isinstance(rbs[0], ResBlock1):      # ResBlock1 on bf16 tensors: every (dilated conv, conv) pair of every branch on the pair kernel,
                for n in range(3):                 # or - wide stages - both convs of the pair on the chunked bf16 tile kernel
                    pair = hipops.resblock1_pairs_ok(B, C, L, [rb.kernel_size for rb in rbs], [rb.convs1[n].dilation for rb in rbs],
                                                     [1] * len(rbs), slope=LRELU_SLOPE)
                    convs = C >= 64 and all(rb.kernel_size >= 3 and
                                            hipops.conv_bf16_config(B, 1, C, C, L, rb.kernel_size, rb.convs1[n].dilation, 1, io_bf16=3) is not None and
                                            hipops.conv_bf16_config(B, 1, C, C, L, rb.kernel_size, 1, 1, io_bf16=3) is not None for rb in rbs)
                    ok = ok and all(rb.kernel_size % 2 == 1 for rb in rbs) and (pair or convs)
            elif C in (16, 32):       # the narrow stages exist as ONE fused kernel only: the library says whether it takes this block set
                ok = ok and all(rb.kernel_size % 2 == 1 for rb in rbs) and hipops.resblock2_stage_split_ok(
                    B, C, L, [rb.kernel_size for rb in rbs], [rb.convs[0].dilation for rb in rbs], [rb.convs[1].dilation for rb in rbs],
                    slope=LRELU_SLOPE)
            else:                   # a wide stage the one-kernel form declines runs conv by conv: every conv needs its bf16 tile kernel
                for rb in rbs:
                    for c in rb.convs:
                        ok = ok and rb.kernel_size % 2 == 1 and rb.kernel_size >= 3 and \
                            hipops.conv_bf16_config(B, 1, C, C, L, rb.kernel_size, c.dilation, 1, io_bf16=3) is not None
        self._fold_key['bf16_storage_ok'] = (key, bool(ok))
        return bool(ok)

    def _inline_stats_ok(self, B, T) -> bool:
        """Statistics without launches need, for EVERY stage, a producer that adds its sums to the accumulator (the stand-alone bf16
        transposed conv of stage 0, the fused upsampler behind every other stage) and a consumer that folds them (the resident-tile
        stage kernels, the 16-channel kernel): asked of the library before the forward starts (shape queries only)."""
        key = (B, T, self.num_kernels, tuple(self.fuse_stage), bool(self.fuse_up), bool(self.fuse_wide_stage), self.fuse_post)
        hit = self._fold_key.get('inline_stats_ok')
        if hit is not None and hit[0] == key:
            return hit[1]
        nk, ns = self.num_kernels, self.num_upsamples
        ok = bool(self.fuse_up and self.fuse_wide_stage) and all(isinstance(rb, ResBlock2) for rb in self.resblocks)
        L = T
        for i, up in enumerate(self.ups):
            if not ok:
                break
            Lo, C = L * up.stride, up.out_channels
            rbs = self.resblocks[i * nk:(i + 1) * nk]
            ks, d1, d2 = [rb.kernel_size for rb in rbs], [rb.convs[0].dilation for rb in rbs], [rb.convs[1].dilation for rb in rbs]
            if i == 0:      # the producer of stage 0: the stand-alone transposed conv (bf16 in, bf16 out)
                ok = ok and hipops.convt_bf16_stats_tiles(_Shape(B, up.in_channels, L), _Shape(B, C, Lo), up.kernel_size, up.stride, io_bf16=3, acc=True) > 0
            if i + 1 < ns:  # the stage kernel folds its input's statistics and runs the next upsampler, which adds up the next stage's
                nup = self.ups[i + 1]
                ok = ok and C >= 32 and (C >= 64 or C in self.fuse_stage) and nup.kernel_size == 2 * nup.stride and nup.stride in (2, 4) \
                    and nup.out_channels * 2 == C and hipops.resblock2_stage_up_tiles(
                        B, C, Lo, ks, d1, d2, slope=LRELU_SLOPE, up_k=nup.kernel_size, up_u=nup.stride, up_slope=LRELU_SLOPE, fold=True) > 0
            else:           # the last stage: any one-kernel form that folds (with the tail behind it when fuse_post takes it)
                ok = ok and hipops.resblock2_stage_split_ok(B, C, Lo, ks, d1, d2, slope=LRELU_SLOPE, fold=True)
            L = Lo
        self._fold_key['inline_stats_ok'] = (key, bool(ok))
        return bool(ok)

    # -------------------------------------------------------------------------------------------
    @_hip.on_tensor_device
    def forward(self, x, spk_emb=None, noise=None):
        """x (B, num_wv_feat, T) channels-first, spk_emb (B, spk_dim), noise (B, noise_dim) -> (B, 1, T*prod(rates))."""
        if spk_emb is None or noise is None:
            # the reference's torch.cat((None, None)) raises TypeError
            raise TypeError('Generator.forward: spk_emb and noise are required')
        if not (x.is_cuda and spk_emb.is_cuda and noise.is_cuda):
            raise RuntimeError('Generator.forward runs on the MI355X HIP path only: move inputs and module to a GPU '
                               '(there is no CPU/PyTorch fallback)')
        dev = x.device
        if self.conv_pre.bias.device != dev:
            raise RuntimeError(f'Generator parameters live on {self.conv_pre.bias.device}, inputs on {dev}')
        if x.dim() != 3 or x.shape[1] != self.h.num_wv_feat:
            raise RuntimeError(f'expected x of shape (B, {self.h.num_wv_feat}, T), got {tuple(x.shape)}')
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError('Generator (HIP): gradients w.r.t. the latent input x are not provided (the reference '
                                      'training loop never asks for them, vec2wav/train.py:154-167)')
        x = x.detach().contiguous().float()
        spk = spk_emb.detach().contiguous().float()
        nz = noise.detach().contiguous().float()
        if spk.shape != (x.shape[0], self.h.spk_dim) or nz.shape != (x.shape[0], self.h.noise_dim):
            raise RuntimeError('spk_emb / noise must be (B, spk_dim) / (B, noise_dim)')
        if needs_grad:
            if self.algo == hipops.ALGO_DIRECT:
                raise NotImplementedError('Generator (HIP): the scalar cross-check kernels (algo = ALGO_DIRECT) serve no-grad forwards only; '
                                          'back-propagation needs the MFMA schedule (algo = ALGO_AUTO)')
            from .backward import GeneratorFunction
            names, params = zip(*[(n, q) for n, q in self.named_parameters()])
            return GeneratorFunction.apply(self, names, x, spk, nz, *params)
        return self._forward_hip(x, spk, nz, None)

    def _forward_hip(self, x, spk, nz, save):
        """The launch schedule of one forward.  `save` (a dict) switches to the back-propagatable form: every intermediate in
        its own fresh buffer (handed over in save['ws']), no stage / pair fusion, plain-layout weights kept next to the packed ones."""
        dev = x.device
        B, _, T = x.shape
        training = self.training
        algo = self.algo
        nk = self.num_kernels
        c0 = self.h.upsample_initial_channel
        keep_ws = None
        if save is not None:
            keep_ws, self._ws = self._ws, {}
        fuse_stage = () if save is not None else self.fuse_stage
        fuse_pairs = () if save is not None else self.fuse_pairs
        # bf16 activation STORAGE (BASELINE configs[2] priced at 2 bytes per activation): every layer of the no-grad bf16 forward reads
        # and writes bf16 tensors when all of them run on the bf16 kernels - default ResBlock2 generator, wide stages C % 32 == 0
        # (>= 64), narrow stages 32 / 16 fused.  fp32 accumulate, fp32 BatchNorm statistics from the accumulators, fp32 output.
        adt = torch.float32
        rb1_net = all(isinstance(rb, ResBlock1) for rb in self.resblocks)
        if (self.precision == 'bf16' and self.bf16_storage and save is None and algo == hipops.ALGO_AUTO and nk <= 3
                and (all(isinstance(rb, ResBlock2) for rb in self.resblocks) or rb1_net) and x.shape[2] % 4 == 0 and self.h.num_wv_feat % 32 == 0
                and all((up.out_channels >= 64 and up.out_channels % 64 == 0) or up.out_channels in fuse_stage for up in self.ups)
                and all(up.in_channels % 32 == 0 and 2 <= up.stride <= 8 for up in self.ups)
                and self._bf16_storage_kernels_exist(B, T)):
            adt = torch.bfloat16
        st = adt == torch.bfloat16

        with torch.no_grad():
            main = torch.cuda.current_stream(dev)
            side = self._side_stream(dev)
            side.wait_stream(main)
            # bf16 storage: the side stream's work is ordered by first use and joined piecewise through events (`need`)
            evs = {} if st else None
            # ---- K3: gamma/beta of every stage (depends on spk/noise only); spectral-norm u/v updated in train mode
            ns = self.num_upsamples
            gbs = [self._buf(f'gb.{i}', (B, 2 * self.cbns[i].num_features), device=dev) for i in range(ns)]
            z_ws = self._buf('z_ws', (ns * B * _Z_CHANNEL,), device=dev)
            sigma_ws = self._buf('sigma_ws', (ns,), device=dev)
            # (depends on spk / noise and the conditioning weights only: three latency-bound launches, ~140 us, that run on the side
            # stream - behind the upsamplers' weight folds - beside the Conv1d weight batch and conv_pre; joined before the first upsampler)
            # eval-mode inference: gamma / beta, the running statistics and the fold into (a, s) of EVERY stage are one launch
            # (v2w_cond_affine_eval) - nothing between (spk, noise) and the affines depends on the activations; sigma = u^T W v depends on
            # the parameters alone and is kept with the fold cache.  (A forward that will be back-propagated keeps gb / z for its backward.)
            eval_fast = not training and save is None
            cond_state = {'done': False, 'affs': None}

            def run_cond():        # on the side stream; bf16 storage: between ups.0's weights and the Conv1d batch (see _split_weights)
                if cond_state['done']:
                    return
                cond_state['done'] = True
                affs = None
                with torch.cuda.stream(side):
                    if eval_fast:
                        sn_p = [q for c in self.cbns for q in (c.layer.weight_orig, c.layer.weight_u, c.layer.weight_v)]
                        skey = (tuple((q.data_ptr(), q._version) for q in sn_p), str(dev))
                        if self._fold_key.get('sigma') != skey:
                            hipops.cond_sigma([c.layer.weight_orig.detach() for c in self.cbns], [c.layer.weight_u for c in self.cbns],
                                              [c.layer.weight_v for c in self.cbns], sigma_ws, training=False)
                            self._fold_key['sigma'] = skey
                        affs = [(self._buf(f'bn.a{i}', (B, self.cbns[i].num_features), device=dev),
                                 self._buf(f'bn.s{i}', (B, self.cbns[i].num_features), device=dev)) for i in range(ns)]
                        hipops.cond_affine_eval(
                            spk, nz, [f.weight.detach() for f in self.fcs], [f.bias.detach() for f in self.fcs],
                            [c.layer.weight_orig.detach() for c in self.cbns], [c.layer.bias.detach() for c in self.cbns], sigma_ws,
                            [c.batch_nrom.running_mean for c in self.cbns], [c.batch_nrom.running_var for c in self.cbns],
                            [c.batch_nrom.eps for c in self.cbns], [q[0] for q in affs], [q[1] for q in affs])
                    else:
                        self._fold_key.pop('sigma', None)        # (sigma_ws is about to hold this forward's own values)
                        hipops.cond_gamma_beta(
                            spk, nz,
                            [f.weight.detach() for f in self.fcs], [f.bias.detach() for f in self.fcs],
                            [c.layer.weight_orig.detach() for c in self.cbns], [c.layer.bias.detach() for c in self.cbns],
                            [c.layer.weight_u for c in self.cbns], [c.layer.weight_v for c in self.cbns],
                            gbs, z_ws, sigma_ws, training)
                cond_state['affs'] = affs
                if evs is not None:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    evs['cond'] = ev

            # (bf16 storage: the only fp32 fold is conv_post's, first read by the last stage - it runs on the side stream, not in front of conv_pre)
            with (torch.cuda.stream(side) if st else contextlib.nullcontext()):
                wf, wp = self._fold_weights(dev, need_wf=save is not None, bf16_only=st)
            if st:
                ev = torch.cuda.Event()
                ev.record(side)
                evs['post'] = ev
            # bf16 storage: the side stream's work is ordered by first use and joined piecewise through events (`need`)
            wps = self._split_weights(dev, all_ups=st, ups_stream=side, events=evs,   # (the fused C = 32 stage is a no-grad schedule: fuse_stage is empty when saving)
                                      between=run_cond if st else None)

            def ck(nm, io=3):   # kernel choice of one Conv1d layer: split-f16 fragments when prepared, else the f32 MFMA stream
                if nm in wps and nm in self._split_wide:
                    if st:       # bf16 storage: io bit 0 = the input tensor is bf16, bit 1 = out / res / addends are bf16
                        return dict(algo=hipops.ALGO_BF16, wps=wps[nm], io_bf16=io)
                    return dict(algo=hipops.ALGO_BF16 if self.precision == 'bf16' else hipops.ALGO_SPLIT, wps=wps[nm])
                if st:
                    raise RuntimeError(f'bf16 storage: layer {nm} has no bf16 kernel (set generator.bf16_storage = False)')
                return dict(algo=algo, wp=wp[nm])

            run_cond()
            affs = cond_state['affs']

            # ---- statistics without launches (csrc/v2w_bnacc.h): one zeroed accumulator per stage, filled by the producing kernel's atomics and
            # folded by the consuming stage kernel; the memset is the side stream's (its events order it before the first producer)
            inline = bool(training and st and self.inline_stats and self.stat_sync is None and save is None and self._inline_stats_ok(B, T))
            accs = None
            if inline:
                offs = [0]
                for c in self.cbns:
                    offs.append(offs[-1] + 4 * c.num_features)
                acc_all = self._buf('bn.acc', (offs[-1],), dtype=torch.int64, device=dev)
                with torch.cuda.stream(side):
                    acc_all.zero_()
                    ev = torch.cuda.Event()
                    ev.record(side)
                    evs['acc'] = ev
                accs = [acc_all[offs[i]:offs[i + 1]] for i in range(ns)]

            def need(*names):     # the main stream waits for exactly these steps of the side stream (bf16 storage; no-op otherwise)
                for nm in names:
                    ev = evs.pop(nm, None) if evs is not None else None
                    if ev is not None:
                        main.wait_event(ev)

            cond_joined = st     # (bf16 storage joins through `need`)

            y = None
            # ---- K1: conv_pre (no activation in front of it)
            cur = self._buf('act.pre', (B, c0, T), dtype=adt, device=dev)
            slab = self._slab(dev)
            self._timed('conv_pre', hipops.conv1d, x, wf['conv_pre'], self.conv_pre.bias.detach(), cur, k=7, dil=1,
                        slope=1.0, splitk_ws=slab, **ck('conv_pre', io=2))        # (the latents arrive as fp32)
            L = T
            up_done = None      # rows of bn.part{i} when the kernel of stage i - 1 already ran ups[i] (fuse_up)
            for i in range(ns):
                up = self.ups[i]
                C = up.out_channels
                Lo = L * up.stride
                # ---- K2: leaky_relu(0.1) -> ConvTranspose1d
                xr = self._buf(f'act.up{i}', (B, C, Lo), dtype=adt, device=dev)
                if st:
                    if up_done is None:
                        need(f'ups.{i}')                       # this upsampler's fragments (the side stream packed them)
                elif not cond_joined and f'ups.{i}' in wps:    # the side stream holds this upsampler's packed weights (bf16 / f16x3 modes)
                    main.wait_stream(side)
                    cond_joined = True
                cbn = self.cbns[i]
                bn = cbn.batch_nrom
                stats = part = None
                nt_stats = 0
                if inline:
                    need('acc')
                elif training:
                    stats = self._buf(f'bn.stats{i}', (2 * C + 1,), dtype=torch.float64, device=dev)
                    # fused statistics: the MFMA transposed conv emits per-tile (sum, sumsq) from its accumulators
                    if up_done is not None:
                        nt_stats = up_done
                    elif f'ups.{i}' in wps:
                        nt_stats = hipops.convt_bf16_stats_tiles(cur, xr, up.kernel_size, up.stride, io_bf16=3 if st else 0)
                    elif algo != hipops.ALGO_DIRECT and wp[f'ups.{i}'] is not None:
                        nt_stats = hipops.convt_stats_tiles(B, up.in_channels, C, L, up.kernel_size, up.stride)
                    if nt_stats:
                        part = self._buf(f'bn.part{i}', (nt_stats * C * 2,), device=dev)
                if up_done is not None:
                    pass        # the previous stage's kernel has written xr (and the partial sums): models.py:128-129 ran fused behind it
                elif inline:
                    self._timed(f'ups.{i}', hipops.convt1d_bf16, cur, wps[f'ups.{i}'], up.bias.detach(), xr, k=up.kernel_size,
                                u=up.stride, slope=LRELU_SLOPE, io_bf16=3, stats_acc=accs[i])
                elif f'ups.{i}' in wps and (nt_stats or not training):
                    self._timed(f'ups.{i}', hipops.convt1d_bf16, cur, wps[f'ups.{i}'], up.bias.detach(), xr, k=up.kernel_size,
                                u=up.stride, slope=LRELU_SLOPE, stats_part=part, io_bf16=3 if st else 0)
                elif st:
                    raise RuntimeError(f'bf16 storage: ups.{i} has no bf16 kernel (set generator.bf16_storage = False)')
                else:
                    self._timed(f'ups.{i}', hipops.convt1d, cur, wf[f'ups.{i}'], up.bias.detach(), xr, k=up.kernel_size,
                                u=up.stride, slope=LRELU_SLOPE, algo=algo, wp=wp[f'ups.{i}'], stats_part=part, splitk_ws=slab)
                up_done = None
                # ---- K4: batch statistics (train) -> [all-reduce] -> folded per-sample affine a, s
                sliced = training and nt_stats >= 1024 and self.stat_sync is None and save is None
                fold = None
                if inline:      # no launch here: the stage kernel below folds the totals (and updates the running statistics) itself
                    fold = dict(acc=accs[i], gb=gbs[i], count=float(B) * float(Lo), eps=bn.eps, momentum=bn.momentum,
                                running_mean=bn.running_mean, running_var=bn.running_var, nbt=bn.num_batches_tracked)
                elif sliced:
                    pass        # (thousands of partial rows, nothing to all-reduce, no backward that reads the array: the two-level form below)
                elif training:
                    if nt_stats:
                        hipops.bn_reduce_partials(part, nt_stats, C, B * Lo, stats)
                    else:
                        pws = self._buf('bn.partial', (2 * max(C, 256) * 64,), dtype=torch.float64, device=dev)
                        hipops.bn_stats(xr, stats, pws)
                    if self.stat_sync is not None:
                        self._timed(f'stat_sync.{i}', self.stat_sync, stats)
                a_t = self._buf(f'bn.a{i}', (B, C), device=dev)
                s_t = self._buf(f'bn.s{i}', (B, C), device=dev)
                need('cond')              # (bf16 storage: gamma / beta - or the eval-mode affines - are the side stream's second step)
                if not cond_joined:       # (fp32: the side stream only carries gamma / beta - joined as late as their first use, which
                    main.wait_stream(side)    # matters at B = 1, where conv_pre is shorter than the conditioning chain)
                    cond_joined = True
                if inline:
                    pass
                elif sliced:
                    sl = self._buf(f'bn.slices{i}', (hipops.BN_SLICES * 2 * C,), dtype=torch.float64, device=dev)
                    hipops.bn_reduce_finalize_slices(part, nt_stats, B * Lo, sl, gbs[i], bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                     a_t, s_t, momentum=bn.momentum, eps=bn.eps)
                elif affs is None:
                    hipops.bn_finalize(stats, gbs[i], bn.running_mean, bn.running_var, bn.num_batches_tracked, a_t, s_t,
                                       training=training, momentum=bn.momentum, eps=bn.eps)
                aff = None if inline else (a_t, s_t)
                # (bf16 storage: the fragments of the residual convs, of the upsampler a fused stage kernel runs, of the tail)
                need('rest', *([f'ups.{i + 1}'] if i + 1 < ns else ['post']))
                # ---- K6/K7: the num_kernels residual blocks read the same x = a*xr + s; their mean is the next input.
                # The branches are independent until the final sum, so conv n of ALL branches goes out as one launch
                # (heaviest kernel size first); the first nk-1 branches end in their own buffers o_j and the last branch's
                # final conv adds them in the reference's order ((r0 + r1) + r2) / nk  (models.py:135-141).
                xs = self._buf(f'act.rb{i}', (B, C, Lo), dtype=adt, device=dev)
                rbs = [self.resblocks[i * nk + j] for j in range(nk)]
                names = [f'resblocks.{i * nk + j}' for j in range(nk)]
                merged = algo != hipops.ALGO_DIRECT and nk <= 3
                if merged:
                    t1s = [self._buf(f'act.t1_{i}_{j}', (B, C, Lo), dtype=adt, device=dev) for j in range(nk)]
                    outs = [self._buf(f'act.o_{i}_{j}', (B, C, Lo), dtype=adt, device=dev) for j in range(nk - 1)] + [xs]
                    heavy_first = sorted(range(nk), key=lambda j: -rbs[j].kernel_size)

                    def launch(tag_sfx, probs):
                        probs = [probs[j] for j in heavy_first if j in probs]
                        tag = '+'.join(f'{names[j]}.{tag_sfx}' for j, _ in probs)
                        self._timed(tag, hipops.conv1d_multi, [pr for _, pr in probs], splitk_ws=slab)

                    def final_kw(j):
                        if j < nk - 1:
                            return {}
                        return dict(add=outs[:nk - 1], out_div=float(nk))

                    # narrow stages (C = 32 / 16): both convs of a pair in ONE kernel, the intermediate stays in LDS
                    fused_pair = C in fuse_pairs and C in (16, 32) and all(wp[f'{nm}.{c}'] is not None for nm in names
                                                                          for c in (('convs.0', 'convs.1') if isinstance(rbs[0], ResBlock2)
                                                                                    else ('convs1.0', 'convs2.0')))

                    def launch_pairs(tag_sfx, probs):
                        """probs: {j: dict}; first nk-1 branches in one launch, the summing branch after them."""
                        done = True
                        for js in ([j for j in heavy_first if j in probs and j < nk - 1], [nk - 1] if nk - 1 in probs else []):
                            if js and done:
                                tag = '+'.join(f'{names[j]}.{tag_sfx}' for j in js)
                                done = self._timed(tag, hipops.resblock_pair_multi, [probs[j] for j in js])
                        return done

                    if isinstance(rbs[0], ResBlock2):
                        ok = False
                        # ---- the stage with the NEXT stage's upsampler behind it in one kernel (bf16 tensors): xs is never written, the
                        # kernel stores act.up{i+1} and the BatchNorm partial sums of stage i + 1
                        if (st and self.fuse_up and i + 1 < ns and C >= 32 and (C >= 64 and self.fuse_wide_stage or C in fuse_stage)
                                and f'ups.{i + 1}' in wps and all(f'{nm}.convs.{c}' in wps for nm in names for c in (0, 1))):
                            nup = self.ups[i + 1]
                            if nup.kernel_size == 2 * nup.stride and nup.stride in (2, 4) and nup.out_channels * 2 == C:
                                ntn = hipops.resblock2_stage_up_tiles(B, C, Lo, [rb.kernel_size for rb in rbs], [rb.convs[0].dilation for rb in rbs],
                                                                      [rb.convs[1].dilation for rb in rbs], slope=LRELU_SLOPE,
                                                                      up_k=nup.kernel_size, up_u=nup.stride, up_slope=LRELU_SLOPE, fold=inline)
                                if ntn:
                                    xr_n = self._buf(f'act.up{i + 1}', (B, nup.out_channels, Lo * nup.stride), dtype=adt, device=dev)
                                    part_n = self._buf(f'bn.part{i + 1}', (ntn * nup.out_channels * 2,), device=dev) if training and not inline else None
                                    ok = self._timed('stage:' + '+'.join(f'{nm}.0&1' for nm in names) + f'+ups.{i + 1}', hipops.resblock2_stage_split,
                                                     xr, aff, [dict(wps1=wps[nm + '.convs.0'], b1=rb.convs[0].bias.detach(),
                                                                    wps2=wps[nm + '.convs.1'], b2=rb.convs[1].bias.detach(), k=rb.kernel_size,
                                                                    dil1=rb.convs[0].dilation, dil2=rb.convs[1].dilation)
                                                               for nm, rb in zip(names, rbs)], None, slope=LRELU_SLOPE, out_div=float(nk),
                                                     bf16=True, io_bf16=3, fold=fold, up_acc=accs[i + 1] if inline else None,
                                                     up=(wps[f'ups.{i + 1}'], nup.bias.detach(), xr_n, part_n, nup.kernel_size, nup.stride, LRELU_SLOPE))
                                    if ok:
                                        up_done = ntn
                        if inline and not ok and i + 1 < ns:
                            raise RuntimeError('inline statistics: the fused stage kernel declined a shape its query accepted')
                        if not ok and C in (16, 32) and C in fuse_stage and all(f'{nm}.convs.{c}' in wps for nm in names for c in (0, 1)):
                            branches = [dict(wps1=wps[nm + '.convs.0'], b1=rb.convs[0].bias.detach(),
                                             wps2=wps[nm + '.convs.1'], b2=rb.convs[1].bias.detach(), k=rb.kernel_size,
                                             dil1=rb.convs[0].dilation, dil2=rb.convs[1].dilation) for nm, rb in zip(names, rbs)]
                            # (the reference's block set and 7-tap tail: the weights-in-registers kernel takes them; any other set / k <= 9 tail
                            # would run on the resident-tile template, measured slower than the two kernels: only with fuse_post = 'any')
                            std_set = [(rb.kernel_size, rb.convs[0].dilation, rb.convs[1].dilation) for rb in rbs] == [(3, 1, 3), (7, 1, 3), (11, 1, 3)]
                            if st and self.fuse_post and i == ns - 1 and C == 16 and self.conv_post.kernel_size <= 9 and \
                                    ((std_set and self.conv_post.kernel_size == 7) or self.fuse_post == 'any'):
                                # the last stage with the generator's tail behind it in ONE kernel (models.py:143-145): the stage's output never
                                # leaves the chip, y is written instead
                                y = torch.empty((B, 1, Lo), device=dev, dtype=torch.float32)
                                ok = self._timed('stage:' + '+'.join(f'{nm}.0&1' for nm in names) + '+conv_post', hipops.resblock2_stage_split,
                                                 xr, aff, branches, None, slope=LRELU_SLOPE, out_div=float(nk), bf16=True, io_bf16=3, fold=fold,
                                                 post=(wf['conv_post'], self.conv_post.bias.detach(), y, self.conv_post.kernel_size, 0.01))
                                if not ok:
                                    y = None
                            if not ok:
                                ok = self._timed('stage:' + '+'.join(f'{nm}.0&1' for nm in names), hipops.resblock2_stage_split, xr, aff,
                                                 branches, xs, slope=LRELU_SLOPE, out_div=float(nk),
                                                 bf16=self.precision == 'bf16', io_bf16=3 if st else 0, fold=fold)
                        if not ok and st and self.fuse_wide_stage and C >= 64 and all(f'{nm}.convs.{c}' in wps for nm in names for c in (0, 1)):
                            # wide stage on bf16 tensors: the WHOLE residual section in one kernel (v2w_stage_bf16_wide.hip): x read once,
                            # t1_j on chip, one fp32 accumulator over the branches
                            ok = self._timed('stage:' + '+'.join(f'{nm}.0&1' for nm in names), hipops.resblock2_stage_split, xr, aff,
                                             [dict(wps1=wps[nm + '.convs.0'], b1=rb.convs[0].bias.detach(),
                                                   wps2=wps[nm + '.convs.1'], b2=rb.convs[1].bias.detach(), k=rb.kernel_size,
                                                   dil1=rb.convs[0].dilation, dil2=rb.convs[1].dilation)
                                              for nm, rb in zip(names, rbs)], xs, slope=LRELU_SLOPE, out_div=float(nk),
                                             bf16=True, io_bf16=3, fold=fold)
                        if inline and not ok:
                            raise RuntimeError('inline statistics: the stage kernel declined a shape its query accepted')
                        if st and not ok and C in (16, 32):
                            raise RuntimeError('bf16 storage: the fused narrow-stage kernel did not take this shape '
                                               '(set generator.bf16_storage = False)')
                        if not ok and C in fuse_stage and all(wp[f'{nm}.convs.{c}'] is not None for nm in names for c in (0, 1)):
                            # the whole residual section of the stage in ONE kernel: x read once, t1_j in LDS, sum in registers
                            ok = self._timed('stage:' + '+'.join(f'{nm}.0&1' for nm in names), hipops.resblock2_stage, xr, aff,
                                             [dict(wp1=wp[nm + '.convs.0'], b1=rb.convs[0].bias.detach(),
                                                   wp2=wp[nm + '.convs.1'], b2=rb.convs[1].bias.detach(), k=rb.kernel_size,
                                                   dil1=rb.convs[0].dilation, dil2=rb.convs[1].dilation)
                                              for nm, rb in zip(names, rbs)], xs, slope=LRELU_SLOPE, out_div=float(nk))
                        if (not ok and C == 8 and 8 in fuse_stage and not st and nk <= 4
                                and all(wf[f'{nm}.convs.{c}'] is not None for nm in names for c in (0, 1))):
                            # 8 channels (the sixth stage of a x640 generator): below every MFMA tile - the whole section as one FMA kernel
                            ok = self._timed('stage:' + '+'.join(f'{nm}.0&1' for nm in names), hipops.resblock2_stage_small, xr, aff,
                                             [dict(wf1=wf[nm + '.convs.0'], b1=rb.convs[0].bias.detach(),
                                                   wf2=wf[nm + '.convs.1'], b2=rb.convs[1].bias.detach(), k=rb.kernel_size,
                                                   dil1=rb.convs[0].dilation, dil2=rb.convs[1].dilation)
                                              for nm, rb in zip(names, rbs)], xs, slope=LRELU_SLOPE, out_div=float(nk))
                        if not ok and fused_pair:
                            ok = launch_pairs('0&1', {j: dict(x=xr, in_affine=aff, wp1=wp[names[j] + '.convs.0'],
                                                              b1=rbs[j].convs[0].bias.detach(), wp2=wp[names[j] + '.convs.1'],
                                                              b2=rbs[j].convs[1].bias.detach(), out=outs[j], k=rbs[j].kernel_size,
                                                              dil1=rbs[j].convs[0].dilation, dil2=rbs[j].convs[1].dilation,
                                                              res_mode=0, slope=LRELU_SLOPE, **final_kw(j)) for j in range(nk)})
                        if not ok and st and self.fuse_wide and C >= 64 and all(f'{nm}.convs.{c}' in wps for nm in names for c in (0, 1)):
                            # wide stage on bf16 tensors: the first convs of all branches in ONE launch (x staged once), then the second
                            # convs in one launch on one accumulator (v2w_branch_convs_bf16_fwd; o_j never written)
                            ks = [rb.kernel_size for rb in rbs]
                            ok = self._timed('bconv0:' + '+'.join(f'{nm}.0' for nm in names), hipops.branch_convs_bf16, 0, [xr], aff,
                                             [wps[nm + '.convs.0'][0] for nm in names], [rb.convs[0].bias.detach() for rb in rbs], t1s,
                                             ks, [rb.convs[0].dilation for rb in rbs], slope=LRELU_SLOPE)
                            if ok:
                                ok2 = self._timed('bconv1:' + '+'.join(f'{nm}.1' for nm in names), hipops.branch_convs_bf16, 1, t1s, None,
                                                  [wps[nm + '.convs.1'][0] for nm in names], [rb.convs[1].bias.detach() for rb in rbs], [xs],
                                                  ks, [rb.convs[1].dilation for rb in rbs], slope=LRELU_SLOPE, out_div=float(nk))
                                if not ok2:
                                    conv2 = {j: (j, (t1s[j], wf[names[j] + '.convs.1'], rbs[j].convs[1].bias.detach(), outs[j],
                                                     dict(k=rbs[j].kernel_size, dil=rbs[j].convs[1].dilation, slope=LRELU_SLOPE,
                                                          res=t1s[j], **ck(names[j] + '.convs.1'), **final_kw(j)))) for j in range(nk)}
                                    if nk > 1:
                                        launch('1', {j: conv2[j] for j in range(nk - 1)})
                                    launch('1', {nk - 1: conv2[nk - 1]})
                        if not ok:
                            launch('0', {j: (j, (xr, wf[names[j] + '.convs.0'], rbs[j].convs[0].bias.detach(), t1s[j],
                                                 dict(k=rbs[j].kernel_size, dil=rbs[j].convs[0].dilation, slope=LRELU_SLOPE,
                                                      in_affine=aff, res=xr, res_affine=aff, **ck(names[j] + '.convs.0')))) for j in range(nk)})
                            conv2 = {j: (j, (t1s[j], wf[names[j] + '.convs.1'], rbs[j].convs[1].bias.detach(), outs[j],
                                             dict(k=rbs[j].kernel_size, dil=rbs[j].convs[1].dilation, slope=LRELU_SLOPE,
                                                  res=t1s[j], **ck(names[j] + '.convs.1'), **final_kw(j))))
                                     for j in range(nk)}
                            if nk > 1:
                                launch('1', {j: conv2[j] for j in range(nk - 1)})
                            launch('1', {nk - 1: conv2[nk - 1]})
                    else:
                        xas = [self._buf(f'act.xa_{i}_{j}', (B, C, Lo), dtype=adt, device=dev) for j in range(nk)]
                        xbs = [self._buf(f'act.xb_{i}_{j}', (B, C, Lo), dtype=adt, device=dev) for j in range(nk)]
                        srcs, src_aff = [xr] * nk, aff
                        for n in range(3):
                            dsts = [xas, xbs, outs][n]
                            if save is not None:   # backward needs every sub-block's conv1 output: one buffer per n
                                t1s = [self._buf(f'act.t1_{i}_{j}_{n}', (B, C, Lo), device=dev) for j in range(nk)]
                            ok = False
                            if st and hipops.resblock1_pairs_ok(B, C, Lo, [rb.kernel_size for rb in rbs], [rb.convs1[n].dilation for rb in rbs],
                                                                [1] * nk, slope=LRELU_SLOPE):
                                # (a pair whose resident tiles do not fit - 256 channels at dilation 5: 94 KB of x beside 66 KB of intermediate -
                                # runs conv by conv on the chunked bf16 kernel below, still on bf16 tensors)
                                # ResBlock1 on bf16 tensors (models.py:37-44): pair n of every branch as one resident-tile launch - the dilated
                                # conv's output stays in LDS, the pair's residual joins the output - and the last pair of the last branch adds
                                # the other branches' results in the reference's order ((r0 + r1) + r2) / nk
                                brs = [dict(wps1=wps[f'{names[j]}.convs1.{n}'], b1=rbs[j].convs1[n].bias.detach(),
                                            wps2=wps[f'{names[j]}.convs2.{n}'], b2=rbs[j].convs2[n].bias.detach(), k=rbs[j].kernel_size,
                                            dil1=rbs[j].convs1[n].dilation, dil2=1) for j in range(nk)]
                                sets = [list(range(nk))] if n < 2 or nk == 1 else [list(range(nk - 1)), [nk - 1]]
                                for js in sets:
                                    last = n == 2 and js[-1] == nk - 1
                                    tag = 'rb1:' + '+'.join(f'{names[j]}.{2 * n}&{2 * n + 1}' for j in js)
                                    ok = self._timed(tag, hipops.resblock1_pairs_bf16, [srcs[j] for j in js], src_aff, [brs[j] for j in js],
                                                     [dsts[j] for j in js], slope=LRELU_SLOPE, out_div=float(nk) if last else 0.0,
                                                     add=outs[:nk - 1] if last and nk > 1 else None)
                                    if not ok:
                                        raise RuntimeError('bf16 storage: the ResBlock1 pair kernel declined a shape its query accepted')
                                srcs, src_aff = dsts, None
                                continue
                            if fused_pair:
                                ok = launch_pairs(f'{2 * n}&{2 * n + 1}',
                                                  {j: dict(x=srcs[j], in_affine=src_aff, wp1=wp[f'{names[j]}.convs1.{n}'],
                                                           b1=rbs[j].convs1[n].bias.detach(), wp2=wp[f'{names[j]}.convs2.{n}'],
                                                           b2=rbs[j].convs2[n].bias.detach(), out=dsts[j], k=rbs[j].kernel_size,
                                                           dil1=rbs[j].convs1[n].dilation, dil2=1, res_mode=1, slope=LRELU_SLOPE,
                                                           **(final_kw(j) if n == 2 else {})) for j in range(nk)})
                            if not ok:
                                launch(str(2 * n), {j: (j, (srcs[j], wf[f'{names[j]}.convs1.{n}'], rbs[j].convs1[n].bias.detach(),
                                                            t1s[j], dict(k=rbs[j].kernel_size, dil=rbs[j].convs1[n].dilation,
                                                                         slope=LRELU_SLOPE, in_affine=src_aff, **ck(f'{names[j]}.convs1.{n}')))) for j in range(nk)})
                                conv2 = {j: (j, (t1s[j], wf[f'{names[j]}.convs2.{n}'], rbs[j].convs2[n].bias.detach(), dsts[j],
                                                 dict(k=rbs[j].kernel_size, dil=1, slope=LRELU_SLOPE, res=srcs[j],
                                                      res_affine=src_aff, **ck(f'{names[j]}.convs2.{n}'),
                                                      **(final_kw(j) if n == 2 else {})))) for j in range(nk)}
                                if n < 2:
                                    launch(str(2 * n + 1), conv2)
                                else:
                                    if nk > 1:
                                        launch('5', {j: conv2[j] for j in range(nk - 1)})
                                    launch('5', {nk - 1: conv2[nk - 1]})
                            srcs, src_aff = dsts, None
                else:
                    t1 = self._buf(f'act.t1_{i}', (B, C, Lo), device=dev)
                    for j in range(nk):
                        rb, name = rbs[j], names[j]
                        k = rb.kernel_size
                        last = dict(accumulate=(j > 0), out_div=(float(nk) if j == nk - 1 else 0.0))
                        if isinstance(rb, ResBlock2):
                            c1, c2 = rb.convs[0], rb.convs[1]
                            self._timed(name + '.0', hipops.conv1d, xr, wf[name + '.convs.0'], c1.bias.detach(), t1, k=k,
                                        dil=c1.dilation, slope=LRELU_SLOPE, in_affine=aff, res=xr, res_affine=aff, splitk_ws=slab, **ck(name + '.convs.0'))
                            self._timed(name + '.1', hipops.conv1d, t1, wf[name + '.convs.1'], c2.bias.detach(), xs, k=k,
                                        dil=c2.dilation, slope=LRELU_SLOPE, res=t1, splitk_ws=slab, **ck(name + '.convs.1'), **last)
                        else:
                            xa = self._buf(f'act.xa_{i}', (B, C, Lo), device=dev)
                            xb = self._buf(f'act.xb_{i}', (B, C, Lo), device=dev)
                            src, src_aff = xr, aff
                            dsts = [xa, xb, xs]
                            for n in range(3):
                                c1, c2 = rb.convs1[n], rb.convs2[n]
                                self._timed(f'{name}.{2 * n}', hipops.conv1d, src, wf[f'{name}.convs1.{n}'], c1.bias.detach(), t1,
                                            k=k, dil=c1.dilation, slope=LRELU_SLOPE, in_affine=src_aff, splitk_ws=slab, **ck(f'{name}.convs1.{n}'))
                                extra = last if n == 2 else {}
                                self._timed(f'{name}.{2 * n + 1}', hipops.conv1d, t1, wf[f'{name}.convs2.{n}'], c2.bias.detach(),
                                            dsts[n], k=k, dil=1, slope=LRELU_SLOPE, res=src, res_affine=src_aff, splitk_ws=slab, **ck(f'{name}.convs2.{n}'), **extra)
                                src, src_aff = dsts[n], None
                cur = xs
                L = Lo
            if not cond_joined or st:
                main.wait_stream(side)        # (bf16 storage: whatever step nobody asked for; the side stream has long finished)
            # ---- K8: leaky_relu(0.01) -> conv_post -> tanh (unless the last stage's kernel has already done it)
            if y is None:
                y = torch.empty((B, 1, L), device=dev, dtype=torch.float32)
                self._timed('conv_post', hipops.conv_post_tanh, cur, wf['conv_post'], self.conv_post.bias.detach(), y, k=7,
                            slope=0.01)

        if save is not None:
            # the spectral-norm vectors AS THIS FORWARD LEFT THEM: the backward of sigma = u^T W v must not see a later forward's
            # power-iteration step (two micro-batches before one backward, a DDP buffer broadcast)
            save['sn_uv'] = [(c.layer.weight_u.detach().clone(), c.layer.weight_v.detach().clone()) for c in self.cbns]
            save.update(ws=self._ws, wf=wf, wp=wp, wpd=self._fold_key.get('wpd', {}), vers=self._fold_key.get('vers'),
                        y=y, x=x, spk=spk, nz=nz, training=training, B=B, T=T)
            self._ws = keep_ws
            self._fold_key.pop('state', None)     # the cached fold pointed into the handed-over buffers
        return y
