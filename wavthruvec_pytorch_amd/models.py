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
        from .backward import ResBlockFunction
        names, params = zip(*rb.named_parameters())
        return ResBlockFunction.apply(rb, pairs, names, x, *params)
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
        self.fuse_bn_finalize = True          # train mode: a stage's statistics reduction and its finalisation as one launch (v2w_bn_reduce_finalize) where nothing is all-reduced in between
        self.merge_waits = True               # bf16 storage: one wait per side stream and call, earlier steps of the stream count as met (forward_plan.need)
        self.cond_stream = None               # bf16 storage, train mode: the conditioning chain on a second side stream, beside the weight folds instead of between them.  None: where it pays - while conv_pre + ups.0 are shorter than the one side stream's chain + folds (measured, tools/exp/cond_stream_sweep.py: -26 / -12 us at B x T = 4 096 / 8 192 frames, +8 at 2 048, +13 .. +28 from 12 288 up); True / False: always / never
        self.fuse_post = True                 # leaky_relu -> conv_post -> tanh inside the kernel of the last (C = 16) stage (bf16 storage, and - round 5 - the fp32 stage kernel): the stage's
                                              # output (335 MB at configs[2]) is never written nor read back (v2w_stage_bf16_n16.hip, 7-tap tail)
        self.fuse_stage_backward = False      # exact fp32 backward of a narrow stage (C in fuse_stage): both input-gradient convs of all branches in
                                              # ONE kernel (v2w_stage_args::bwd_*, round 5) instead of three merged launches.  Off: measured - the two
                                              # kernels take 0.2 ms less than the six launches they replace, the step 0.35 ms MORE
        self._split_wide = set()
        self._ws: Dict[str, torch.Tensor] = {}
        self._slabs: Dict[tuple, 'hipops.SplitKSlab'] = {}
        self._wts: Dict[str, torch.Tensor] = {}
        self._fold_key: Dict[str, tuple] = {}
        self._profile = None                  # list -> (tag, start_event, end_event) per conv launch (bench.py roofline)
        self._profile_names = None            # dict -> tag: kernel names, filled beside `_profile` by profile_kernel_names
        self.use_launch_plan = True           # no-grad forwards: planned once per configuration, replayed from the recorded tape (schedule.py)
        self._tapes: Dict[tuple, object] = {}
        self._ws_epoch = 0                    # bumped whenever a workspace / weight buffer is (re)allocated: recorded tapes point into them
        self._tape_refused = 0                # plans not kept because a recorded call pointed outside the module's own memory (schedule.TapeNotOwned)

    # -------------------------------------------------------------------------------------------
    def __getstate__(self):
        """copy.deepcopy / pickle / torch.save of the module (the EMA or snapshot pattern): parameters, buffers and switches travel; recorded
        launch plans (ctypes structs and function pointers into THIS module's buffers), workspaces, folded weights, split-over-C_in slabs
        and the profile list do not - the copy plans, allocates and folds for itself at its first forward."""
        st = self.__dict__.copy()
        st.update(_tapes={}, _ws={}, _slabs={}, _wts={}, _fold_key={}, _profile=None, _profile_names=None, _ws_epoch=0)
        return st

    def enable_sync_batchnorm(self, group=None, single_rank_collective=None):
        """Data-parallel CondBN: all-reduce the per-stage batch statistics over `group` (RCCL on GPUs).  A one-rank group exchanges
        nothing unless `single_rank_collective=True` (bench.py --force-pg and the -m gpu tests: the RCCL code path on a one-GPU box)."""
        from .distributed import BNStatSync
        self.stat_sync = BNStatSync(group, single_rank_collective=single_rank_collective)
        return self

    def profile_kernel_names(self, x, spk_emb, noise):
        """tag -> names of the kernels behind each tagged launch of ONE no-grad forward of these inputs (the tags of `_profile`), as
        `rocprofv3 --kernel-trace` prints them minus namespace and parameter list.  Asked of the library (v2w_name_sink: every launching entry
        point reports the kernels it selects), never reconstructed from tile tables on this side.  The forward runs; its timing is not meant
        to be used (the name queries sit between its launches)."""
        keep = self._profile
        self._profile, self._profile_names = [], {}
        try:
            with torch.no_grad():
                self.forward(x, spk_emb, noise)
            return dict(self._profile_names)
        finally:
            self._profile, self._profile_names = keep, None

    def capture_graph(self, x, spk_emb, noise, warmup: int = 2):
        """Capture one forward (current train/eval mode, these shapes) into a HIP graph and return `run(x, spk_emb, noise)`.

        The forward is ~40-60 kernel launches; at inference sizes (B=1, T~50) it is launch-bound, and replaying one graph
        removes the per-launch host cost.  Inputs are copied into static buffers, the returned tensor is the graph's static
        output (clone it to keep it across calls).  Data-parallel statistics exchange cannot be captured.
        Every buffer the captured launches touch - activations, folded weights, the split-over-C_in scratch (`_slab`) - belongs to
        THIS module, so graphs of different generators may replay concurrently on different streams.  One captured configuration per module
        at a time: a later forward of another shape replaces those buffers, and `run` then refuses to replay (outgrown split-over-C_in slabs
        are kept alive, hipops.SplitKSlab.retired)."""
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

        epoch = self._ws_epoch

        def run(x, spk_emb, noise):
            if self._ws_epoch != epoch:
                # the captured launches write into module-owned buffers: a forward of another shape / precision / storage has replaced them
                raise RuntimeError('capture_graph: the module reallocated its buffers since this graph was captured (a forward of another '
                                   'configuration ran): capture again - a module serves one captured configuration at a time')
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
    def _side_stream(device, index=0):
        key = str(device) if index == 0 else f'{device}#{index}'
        st = _SIDE_STREAMS.get(key)              # per process and device, not per module: modules stay deep-copyable / picklable
        if st is None:
            st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        return st

    def _buf(self, name, shape, dtype=torch.float32, device=None):
        t = self._ws.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device:
            if t is not None:
                self._drop_tapes()
            t = torch.empty(shape, device=device, dtype=dtype)
            self._ws[name] = t
        return t

    def _drop_tapes(self):
        """A buffer recorded launch plans point into is about to go away: forget the plans (the next forward of each configuration plans again)."""
        self._ws_epoch += 1
        self._tapes.clear()

    def _slab(self, device):
        """This module's split-over-C_in scratch for launches on the CURRENT stream of `device` (hipops.SplitKSlab; v2w_conv1d_args::splitk_ws).
        One slab per (module, stream): two generators - or one generator on two streams - never share one, so captured graphs of
        different modules may replay concurrently."""
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        slab = self._slabs.get(key)
        if slab is None:
            if len(self._slabs) >= 16:        # (a module that has launched on many streams: the oldest slab goes, with the plans that name it)
                self._slabs.pop(next(iter(self._slabs)))
                self._drop_tapes()
            slab = self._slabs[key] = hipops.SplitKSlab()
        return slab

    def _wbuf(self, name, shape, dtype=torch.float32, device=None):
        """Folded / packed WEIGHT buffers: like `_buf`, but they stay with the module when a training forward hands its activation
        buffers to the autograd graph - the batched fold's descriptor table (pointers) then survives from step to step instead of being
        rebuilt and copied to the device in front of every training forward.  A backward whose forward's weights have since been
        modified AND re-folded by a later forward is refused (backward.py), as autograd refuses an in-place modified saved tensor."""
        t = self._wts.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device:
            if t is not None:
                self._drop_tapes()
            t = torch.empty(shape, device=device, dtype=dtype)
            self._wts[name] = t
        return t

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
                self._drop_tapes()            # (recorded launch plans point at the old plan's descriptor table)
                plan = hipops.FoldPlan(batch, device)
                plan.key = key
                self._fold_key['plan'] = plan
            plan.run()
        self._fold_key.update(state=state, wf=wf, wp=wp, wpd=wpd, vers=tuple(vers), gen=self._fold_key.get('gen', 0) + 1)
        return wf, wp

    def _split_weights(self, device, all_ups=False, ups_stream=None, mark=None, between=None):
        """precision == 'f16x3': (hi, lo) half-precision fragments + scale record of every Conv1d layer the split kernel
        serves.  Follows the fold cache: rebuilt whenever `_fold_weights` rebuilt (train mode: every forward).
        `ups_stream`: the upsamplers' fold + pack launches (ten latency-bound kernels, ~55 us at five stages) go to that stream - the
        caller joins it before the first upsampler; they then run beside the Conv1d batch and conv_pre.
        `mark` (the planner's: records an event on the side stream under a name) + `between` (a callable): the order on `ups_stream` becomes
        ups.0 -> between() -> ups.1 .. ups.n -> every Conv1d but conv_pre, with an event behind ups.0 ('ups.0') and behind the last step ('rest': it stands for the
        upsamplers in front of it too): the caller waits for exactly what its next launch
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
        if mark is None or ups_stream is None:
            def mark(name):
                pass

        def run_plan(slot, sub, strm):
            if not sub:
                return
            key = tuple((v.data_ptr(), 0 if g is None else g.data_ptr(), w.data_ptr()) for (v, g, w, _s) in sub) + (self.precision,)
            plan = self._fold_key.get(slot)
            if plan is None or plan.key != key:
                self._drop_tapes()
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
                if self._ws.get(f'wpsbf.ups.{i}') is not w:
                    self._drop_tapes()
                self._ws[f'wpsbf.ups.{i}'] = w
                out[f'ups.{i}'] = w
            if i == 0:           # (the later upsamplers sit in front of `rest`, whose event stands for them: an event record between two
                mark(f'ups.{i}')   # folds costs the side stream 6 - 7 us, five of them made `rest` late for the first stage kernel)

        # the transposed convs run on the bf16 matrix pipe too (v2w_convt1d_bf16_fwd); the narrow upsamplers are memory-side: with fp32
        # tensors the f32 kernel's epilogue moves their bytes faster; with bf16 storage every upsampler runs there
        ups_bf = [(i, m) for i, m in enumerate(self.ups) if self.precision == 'bf16' and (m.out_channels >= 64 or all_ups)]
        run_plan('split_plan', first, None)
        if ups_bf and ups_bf[0][0] == 0:       # what the forward needs first: conv_pre's fragments (above, on its own stream), then ups.0's
            fold_up(*ups_bf.pop(0))
        if between is not None:
            between()
        # the other upsamplers' folds (ten latency-bound launches on small layers) in FRONT of the Conv1d batch: the event behind `rest` then stands
        # for every fold of the stream and the first stage's wait is the last one of the forward (round 6: -8 us at B = 32 x T = 256, -3 at
        # B = 64 x T = 512 against folding them behind it, tools/exp/ups_first_ab.py)
        for i, m in ups_bf:
            fold_up(i, m)
        run_plan('split_plan_rest', rest, ups_stream)
        mark('rest')
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
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        x_in = x.contiguous().float()         # (keeps x's place in the autograd graph when it asks for a gradient)
        x = x_in.detach()
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
            return GeneratorFunction.apply(self, names, x_in if x_in.requires_grad else x, spk, nz, *params)
        return self._forward_hip(x, spk, nz, None)

    def _plan_key(self, x):
        """What a recorded launch plan depends on besides the values of the inputs: shape, mode, every switch the planner reads, and the
        storage of every parameter and buffer (a `.to()`, a `load_state_dict(assign=True)` or a replaced Parameter moves them)."""
        ptrs = tuple(p.data_ptr() for p in self.parameters()) + tuple(b.data_ptr() for b in self.buffers())
        return (tuple(x.shape), str(x.device), torch.cuda.current_stream(x.device).cuda_stream, self.training, self.precision, self.algo, self.bf16_storage, tuple(self.fuse_stage), tuple(self.fuse_pairs),
                self.fuse_wide, self.fuse_wide_stage, self.fuse_up, self.fuse_post, self.fuse_bn_finalize, self.cond_stream, self.merge_waits, self.split_min_channels,
                self.always_refold, ptrs)

    def _forward_hip(self, x, spk, nz, save):
        """One forward through the C ABI.  `save` (a dict) asks for the back-propagatable form (forward_plan.py).  A no-grad forward is PLANNED
        once per configuration - forward_plan.ForwardPlanner decides and launches it under a schedule.Recorder - and replayed from the tape
        afterwards: prebuilt argument structs, three input pointers and the output rebound (`use_launch_plan = False`: plan every forward)."""
        from .forward_plan import ForwardPlanner, DirectStreams
        from . import schedule
        dev = x.device
        main, side, side2 = torch.cuda.current_stream(dev), self._side_stream(dev), self._side_stream(dev, 1)
        if save is not None or not self.use_launch_plan or self.stat_sync is not None:
            return ForwardPlanner(self, x, spk, nz, save, DirectStreams(main, side, side2)).run()
        key, refold = self._tape_key(x)
        tape = self._tapes.get(key)
        if tape is not None:
            y = torch.empty(tape.out[0], device=dev, dtype=tape.out[1])
            tape.replay(main, side, dict(x=x.data_ptr(), spk=spk.data_ptr(), nz=nz.data_ptr(), y=y.data_ptr()), self._profile, side2)
            if self._profile_names is not None:
                self._profile_names.update(tape.kernel_names())
            if refold is not None:                      # the replay folded what was stale: what the fold cache must say now
                wvers, sigma = refold
                if wvers is not None:                   # it held the weight folds
                    self._fold_key['gen'] = self._fold_key.get('gen', 0) + 1
                    if wvers:
                        self._fold_key.update(vers=wvers, state=(wvers,) + tape.fold_state_tail)
                if sigma:                               # ... cond_sigma (eval mode)
                    self._fold_key['sigma'] = self._sigma_key(dev)
            if self.training:
                self._fold_key.pop('sigma', None)       # (sigma_ws now holds this forward's own power-iteration values, as the planned forward notes)
            return y
        if self._profile is not None:                   # (a profiled forward of a configuration without a plan yet: planned, timed, not recorded)
            return ForwardPlanner(self, x, spk, nz, None, DirectStreams(main, side, side2)).run()
        rec = schedule.Recorder(_hip.load(), main, side, side2)
        epoch = self._ws_epoch
        prev = _hip.set_recorder(rec)
        try:
            y = ForwardPlanner(self, x, spk, nz, None, rec).run()
        finally:
            _hip.set_recorder(prev)
        binds = dict(x=x.data_ptr(), spk=spk.data_ptr(), nz=nz.data_ptr(), y=y.data_ptr())
        if self._ws_epoch != epoch:                     # buffers were (re)allocated while recording: older tapes point into freed memory
            self._tapes.clear()
        state = self._fold_key.get('state')
        if len(set(binds.values())) == 4 and state is not None:     # (aliased inputs: no tape, the next forward plans again)
            try:
                rec.tape.finalize(binds, owned=self._owned_ranges())
            except schedule.TapeNotOwned:
                self._tape_refused += 1                 # a recorded call points at memory this module does not keep alive: never replay it
                return y
            rec.tape.out = (tuple(y.shape), y.dtype)
            rec.tape.fold_state_tail = tuple(state[1:])
            self._tapes[key] = rec.tape
        return y

    def _owned_ranges(self):
        """Address ranges of everything a recorded launch plan may point into: parameters, buffers, workspaces, folded weights, the
        split-over-C_in slabs (retired ones too) and the tensors of the cached fold / split plans."""
        from . import schedule
        ts = list(self.parameters()) + list(self.buffers()) + list(self._ws.values()) + list(self._wts.values())
        for slab in self._slabs.values():
            ts += [slab.t] + list(slab.retired)
        ts += list(schedule.tensors_in(self._fold_key))
        return schedule.owned_ranges(ts)

    def _sigma_key(self, dev):
        sn_p = [q for c in self.cbns for q in (c.layer.weight_orig, c.layer.weight_u, c.layer.weight_v)]
        return (tuple((q.data_ptr(), q._version) for q in sn_p), str(dev))

    def _tape_key(self, x):
        """(key of the launch plan this forward needs, what that plan refreshes).  A recorded tape holds exactly the launches that were due
        when it was recorded, so the two things that can be stale are SEPARATE parts of the key: 'w' - the weight folds (a conv parameter
        changed), 's' - sigma of the spectral norm (eval mode: a cbns parameter changed, or a train-mode forward left its own values in
        sigma_ws).  Second value: None when the plan refreshes nothing, else (parameter versions the fold cache holds after the folds ran |
        None when the plan has no weight fold - () in train mode with always_refold, where the cache is not consulted -, sigma refreshed)."""
        base = self._plan_key(x)
        if self.training and self.always_refold:
            return base + ('refold',), ((), False)
        vers = self._param_versions()
        wstale = not (self._fold_key.get('vers') == vers and self._fold_key.get('state') is not None)
        sstale = (not self.training) and self._fold_key.get('sigma') != self._sigma_key(x.device)
        if not (wstale or sstale):
            return base + ('folded',), None
        return base + (('w' if wstale else '') + ('s' if sstale else ''),), (vers if wstale else None, sstale)

    def _param_versions(self):
        vers = []
        for _name, m in self._conv_layers():
            ps = (m.weight_v, m.weight_g) if m.weight_normed else (m.weight,)
            vers.append(tuple((p.data_ptr(), p._version) for p in ps))
        return tuple(vers)
