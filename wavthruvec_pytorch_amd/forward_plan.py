"""The planner of `Generator.forward` (reference: vec2wav/models.py:116-147): which C-ABI call serves which step, in which order, on which
stream.  One `ForwardPlanner` plans - and, through the hipops wrappers, launches - ONE forward; under a `schedule.Recorder` the launches are
taped as they go out and every later forward of the same configuration replays the tape (schedule.py).

    begin            bf16 activation storage or fp32 tensors (asked of the library's shape queries), the side stream forked
    weights          weight-norm folds, fragment packs, the conditioning chain: side-stream work in order of first use, an event per step
    conv_pre         models.py:123
    stage(i)         upsample (models.py:128-129) -> statistics (modules.py:23: batch sums, [all-reduce], folded affine) -> residual
                     (models.py:133-141: the first candidate that takes the stage runs it - one kernel with the next upsampler behind it, one
                     kernel, merged branch launches, layer by layer)
    tail             models.py:143-145 unless the last stage's kernel already ran it

A forward that will be back-propagated (`save`) gets every intermediate in a fresh buffer and no stage fusion (backward.py reads them)."""
from __future__ import annotations

import contextlib

import torch

from . import hipops

LRELU_SLOPE = 0.1  # models.py:10
_Z_CHANNEL = 128   # models.py:110


class DirectStreams:
    """The stream operations of a plan, executed and not taped (schedule.Recorder offers the same five and tapes them)."""

    def __init__(self, main, side, side2=None):
        self.main, self.side, self.side2 = main, side, side if side2 is None else side2
        self.events = {}
        self.nfork = 1
        self.tag = None

    def stream(self, sid):
        return (self.main, self.side, self.side2)[sid]

    def fork(self, sides=1):
        self.nfork = sides
        for q in (self.side, self.side2)[:sides]:
            q.wait_stream(self.main)

    def join(self):
        for q in (self.side, self.side2)[:self.nfork]:
            self.main.wait_stream(q)

    def mark(self, name, sid=1):
        ev = torch.cuda.Event()
        ev.record(self.stream(sid))
        self.events[name] = ev

    def need(self, name):
        self.main.wait_event(self.events[name])

    def py(self, fn, tag=None):
        return fn()

    def keep(self, obj):
        pass


def _branches(rbs, names, wps):
    return [dict(wps1=wps[nm + '.convs.0'], b1=rb.convs[0].bias.detach(), wps2=wps[nm + '.convs.1'], b2=rb.convs[1].bias.detach(),
                 k=rb.kernel_size, dil1=rb.convs[0].dilation, dil2=rb.convs[1].dilation) for nm, rb in zip(names, rbs)]


class ForwardPlanner:
    def __init__(self, g, x, spk, nz, save, S):
        self.g, self.x, self.spk, self.nz, self.save, self.S = g, x, spk, nz, save, S
        self.dev = x.device
        self.B, _, self.T = x.shape
        self.training = g.training
        self.algo = g.algo
        self.nk, self.ns = g.num_kernels, g.num_upsamples
        self.marked = set()        # side-stream steps with an event behind them ...
        self.order = []            # ... (name, stream) in the order they were recorded
        self.needed = set()        # ... and the ones the main stream has already waited for

    # ---------------------------------------------------------------------------------------------------------------------------
    def run(self):
        g = self.g
        keep_ws = None
        if self.save is not None:
            keep_ws, g._ws = g._ws, {}
        try:
            with torch.no_grad():
                self.begin()
                self.weights()
                self.conv_pre()
                for i in range(self.ns):
                    self.stage(i)
                y = self.tail()
            if self.save is not None:
                self.hand_over(y)
        finally:
            if keep_ws is not None:
                g._ws = keep_ws
        return y

    def timed(self, tag, fn, *args, **kw):
        """Launch `fn` under `tag`: bench.py's roofline brackets the launches of a tag with events on the launching stream."""
        g = self.g
        self.S.tag = tag
        try:
            if g._profile is None or not isinstance(self.S, DirectStreams):
                return fn(*args, **kw)          # (a recorded plan is profiled at replay)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pn = g._profile_names                # Generator.profile_kernel_names: also ask every launch of the tag for its kernel names
            e0.record()
            if pn is not None:
                from . import _hip, schedule
                prev = _hip.set_recorder(schedule.NameProbe(_hip.load(), pn.setdefault(tag, [])))
                try:
                    r = fn(*args, **kw)
                finally:
                    _hip.set_recorder(prev)
            else:
                r = fn(*args, **kw)
            e1.record()
            g._profile.append((tag, e0, e1))
            return r
        finally:
            self.S.tag = None

    def on_side(self, sid=1):
        return torch.cuda.stream(self.S.stream(sid))

    def mark(self, name, sid=1):
        self.S.mark(name, sid)
        self.marked.add(name)
        self.order.append((name, sid))

    def need(self, *names):
        """The main stream waits for these steps of the side streams - for nothing more, and for each stream at most once per call: the event
        behind a step of an in-order stream stands for every earlier step of that stream (`post` is behind `ups.0`, `rest` behind `ups.1`), so
        the latest of `names` on a stream is the one waited for and the earlier ones count as met (a wait costs the main queue ~3 us even when
        long satisfied: eight per forward before, five now)."""
        want = [nm for nm in names if nm in self.marked and nm not in self.needed]
        if not self.g.merge_waits:
            for nm in want:
                self.needed.add(nm)
                self.S.need(nm)
            return
        for sid in sorted({s for n, s in self.order if n in want}):
            seq = [n for n, s in self.order if s == sid]
            last = max(seq.index(n) for n in want if n in seq)
            self.S.need(seq[last])
            self.needed.update(seq[:last + 1])

    def join_side(self):
        if not self.joined:
            self.S.join()
            self.joined = True

    # ---------------------------------------------------------------------------------------------------------------------------
    def begin(self):
        """bf16 activation STORAGE (BASELINE configs[2] priced at 2 bytes per activation): every layer of the no-grad bf16 forward reads and
        writes bf16 tensors when all of them run on the bf16 kernels - wide stages C % 64 == 0, narrow stages 32 / 16 fused.  fp32 accumulate,
        fp32 BatchNorm statistics from the accumulators, fp32 output."""
        g, x = self.g, self.x
        from .models import ResBlock1, ResBlock2
        self.fuse_stage = () if self.save is not None else g.fuse_stage
        self.fuse_pairs = () if self.save is not None else g.fuse_pairs
        self.adt = torch.float32
        rb1_net = all(isinstance(rb, ResBlock1) for rb in g.resblocks)
        if (g.precision == 'bf16' and g.bf16_storage and self.save is None and self.algo == hipops.ALGO_AUTO and self.nk <= 3
                and (all(isinstance(rb, ResBlock2) for rb in g.resblocks) or rb1_net) and x.shape[2] % 4 == 0 and g.h.num_wv_feat % 32 == 0
                and all((up.out_channels >= 64 and up.out_channels % 64 == 0) or up.out_channels in self.fuse_stage for up in g.ups)
                and all(up.in_channels % 32 == 0 and 2 <= up.stride <= 8 for up in g.ups)
                and g._bf16_storage_kernels_exist(self.B, self.T)):
            self.adt = torch.bfloat16
        self.st = self.adt == torch.bfloat16
        # the conditioning chain (fcs -> spectral-norm step -> gamma / beta: three latency-bound launches, 70 - 125 us beside conv_pre) on a queue of
        # its own in the bf16-storage train-mode plan: behind it on ONE side stream the residual convs' fragments reached the first stage kernel
        # 35 us late (B = 32 x T = 256, round 6 trace)
        cs = g.cond_stream if g.cond_stream is not None else 3072 <= self.B * self.T <= 10240
        self.cond_sid = 2 if (self.st and self.training and cs) else 1
        self.joined = False
        self.S.fork(self.cond_sid)
        self.slab = g._slab(self.dev)
        self.y = None
        self.wps32 = {}
        self.up_done = None        # rows of bn.part{i} when the kernel of stage i - 1 already ran ups[i] (fuse_up)

    def buf(self, name, shape, dtype=torch.float32):
        return self.g._buf(name, shape, dtype=dtype, device=self.dev)

    # ---------------------------------------------------------------------------------------------------------------------------
    def weights(self):
        """K0 + K3.  bf16 storage: the side stream's work - conv_post's fold, ups.0's fragments, the conditioning chain (train mode at mid sizes:
        on the second side stream), the other upsamplers' fragments, the Conv1d batch - with an event behind ups.0, the conditioning chain and
        the last step, and the main stream waits for exactly the event its next launch needs (`need`): three waits per forward.  fp32: the folds run in front of conv_pre (8.55 against 8.65 ms with them beside it), the conditioning chain on the
        side stream, joined as late as its first use."""
        g, st, dev = self.g, self.st, self.dev
        B, ns = self.B, self.ns
        self.gbs = [self.buf(f'gb.{i}', (B, 2 * g.cbns[i].num_features)) for i in range(ns)]
        self.z_ws = self.buf('z_ws', (ns * B * _Z_CHANNEL,))
        self.sigma_ws = self.buf('sigma_ws', (ns,))
        self.affs = None
        self.cond_ran = False
        if self.cond_sid == 2:
            self.cond()                # its own queue: issued first, it runs beside everything below
        with (self.on_side() if st else contextlib.nullcontext()):
            self.wf, self.wp = g._fold_weights(dev, need_wf=self.save is not None, bf16_only=st)
        # (no event of its own behind conv_post's fold: it is the first thing on the side stream, the event behind ups.0 stands for it)
        self.wps = g._split_weights(dev, all_ups=st, ups_stream=self.S.side, mark=self.mark if st else None,
                                    between=self.cond if st else None)
        self.cond()

    def cond(self):
        """gamma / beta of every stage (depend on spk / noise only); spectral-norm u / v updated in train mode.  Eval-mode inference: gamma /
        beta, the running statistics and the fold into (a, s) of EVERY stage are one launch (v2w_cond_affine_eval); sigma = u^T W v depends
        on the parameters alone and is kept with the fold cache."""
        if self.cond_ran:
            return
        self.cond_ran = True
        g, dev, ns, B = self.g, self.dev, self.ns, self.B
        with self.on_side(self.cond_sid):
            if not self.training and self.save is None:
                sn_p = [q for c in g.cbns for q in (c.layer.weight_orig, c.layer.weight_u, c.layer.weight_v)]
                skey = (tuple((q.data_ptr(), q._version) for q in sn_p), str(dev))
                if g._fold_key.get('sigma') != skey:
                    hipops.cond_sigma([c.layer.weight_orig.detach() for c in g.cbns], [c.layer.weight_u for c in g.cbns],
                                      [c.layer.weight_v for c in g.cbns], self.sigma_ws, training=False)
                    g._fold_key['sigma'] = skey
                self.affs = [(self.buf(f'bn.a{i}', (B, g.cbns[i].num_features)), self.buf(f'bn.s{i}', (B, g.cbns[i].num_features))) for i in range(ns)]
                hipops.cond_affine_eval(
                    self.spk, self.nz, [f.weight.detach() for f in g.fcs], [f.bias.detach() for f in g.fcs],
                    [c.layer.weight_orig.detach() for c in g.cbns], [c.layer.bias.detach() for c in g.cbns], self.sigma_ws,
                    [c.batch_nrom.running_mean for c in g.cbns], [c.batch_nrom.running_var for c in g.cbns],
                    [c.batch_nrom.eps for c in g.cbns], [q[0] for q in self.affs], [q[1] for q in self.affs])
            else:
                g._fold_key.pop('sigma', None)        # (sigma_ws is about to hold this forward's own values)
                hipops.cond_gamma_beta(
                    self.spk, self.nz, [f.weight.detach() for f in g.fcs], [f.bias.detach() for f in g.fcs],
                    [c.layer.weight_orig.detach() for c in g.cbns], [c.layer.bias.detach() for c in g.cbns],
                    [c.layer.weight_u for c in g.cbns], [c.layer.weight_v for c in g.cbns], self.gbs, self.z_ws, self.sigma_ws, self.training)
        if self.st:
            self.mark('cond', self.cond_sid)

    def ck(self, nm, io=3):
        """Kernel choice of one Conv1d layer: bf16 / split-f16 fragments when prepared, else the f32 MFMA stream."""
        g = self.g
        if nm in self.wps32:
            return dict(algo=hipops.ALGO_BF16, wps=self.wps32[nm])
        if nm in self.wps and nm in g._split_wide:
            if self.st:       # bf16 storage: io bit 0 = the input tensor is bf16, bit 1 = out / res / addends are bf16
                return dict(algo=hipops.ALGO_BF16, wps=self.wps[nm], io_bf16=io)
            return dict(algo=hipops.ALGO_BF16 if g.precision == 'bf16' else hipops.ALGO_SPLIT, wps=self.wps[nm])
        if self.st:
            raise RuntimeError(f'bf16 storage: layer {nm} has no bf16 kernel (set generator.bf16_storage = False)')
        return dict(algo=self.algo, wp=self.wp[nm])

    # ---------------------------------------------------------------------------------------------------------------------------
    def conv_pre(self):
        g = self.g
        self.cur = self.buf('act.pre', (self.B, g.h.upsample_initial_channel, self.T), dtype=self.adt)
        self.timed('conv_pre', hipops.conv1d, self.x, self.wf['conv_pre'], g.conv_pre.bias.detach(), self.cur, k=7, dil=1,
                   slope=1.0, splitk_ws=self.slab, **self.ck('conv_pre', io=2))        # (the latents arrive as fp32)
        self.L = self.T

    def stage(self, i):
        up = self.g.ups[i]
        self.C, self.Lo = up.out_channels, self.L * up.stride
        self.xr = self.buf(f'act.up{i}', (self.B, self.C, self.Lo), dtype=self.adt)
        nt_stats, part = self.upsample(i)
        self.aff = self.statistics(i, nt_stats, part)
        # (bf16 storage: the fragments of the residual convs, of the upsampler a fused stage kernel runs, of the tail)
        self.need('rest', f'ups.{i + 1}' if i + 1 < self.ns else 'post')
        self.xs = self.buf(f'act.rb{i}', (self.B, self.C, self.Lo), dtype=self.adt)
        self.residual(i)
        self.cur, self.L = self.xs, self.Lo

    def upsample(self, i):
        """K2: leaky_relu(0.1) -> ConvTranspose1d (+ the per-tile BatchNorm sums of its output)."""
        g, st, up, C = self.g, self.st, self.g.ups[i], self.C
        B, L, dev = self.B, self.L, self.dev
        if st:
            if self.up_done is None:
                self.need(f'ups.{i}')                       # this upsampler's fragments (the side stream packed them)
        elif f'ups.{i}' in self.wps:                        # the side stream holds this upsampler's packed weights (bf16 / f16x3 modes)
            self.join_side()
        part, nt_stats = None, 0
        if self.training:
            # fused statistics: the MFMA transposed conv emits per-tile (sum, sumsq) from its accumulators
            if self.up_done is not None:
                nt_stats = self.up_done
            elif f'ups.{i}' in self.wps:
                nt_stats = hipops.convt_bf16_stats_tiles(self.cur, self.xr, up.kernel_size, up.stride, io_bf16=3 if st else 0)
            elif self.algo != hipops.ALGO_DIRECT and self.wp[f'ups.{i}'] is not None:
                nt_stats = hipops.convt_stats_tiles(B, up.in_channels, C, L, up.kernel_size, up.stride)
            if nt_stats:
                part = self.buf(f'bn.part{i}', (nt_stats * C * 2,))
        if self.up_done is not None:
            pass        # the previous stage's kernel has written xr (and the partial sums): models.py:128-129 ran fused behind it
        elif f'ups.{i}' in self.wps and (nt_stats or not self.training):
            self.timed(f'ups.{i}', hipops.convt1d_bf16, self.cur, self.wps[f'ups.{i}'], up.bias.detach(), self.xr, k=up.kernel_size,
                       u=up.stride, slope=LRELU_SLOPE, stats_part=part, io_bf16=3 if st else 0)
        elif st:
            raise RuntimeError(f'bf16 storage: ups.{i} has no bf16 kernel (set generator.bf16_storage = False)')
        else:
            self.timed(f'ups.{i}', hipops.convt1d, self.cur, self.wf[f'ups.{i}'], up.bias.detach(), self.xr, k=up.kernel_size,
                       u=up.stride, slope=LRELU_SLOPE, algo=self.algo, wp=self.wp[f'ups.{i}'], stats_part=part, splitk_ws=self.slab)
        self.up_done = None
        return nt_stats, part

    def statistics(self, i, nt_stats, part):
        """K4: batch statistics (train) -> [all-reduce] -> folded per-sample affine a, s (modules.py:20-30)."""
        g, C, B, Lo = self.g, self.C, self.B, self.Lo
        bn = g.cbns[i].batch_nrom
        # many thousands of partial rows, nothing to all-reduce, no backward that reads the array: the two-level form (below that the one-level
        # kernels keep four row loads in flight: one launch)
        sliced = self.training and nt_stats >= 4096 and g.stat_sync is None and self.save is None
        stats = None
        # ... hundreds of rows, nothing to all-reduce: the reduction and the finalisation as one launch
        fused = g.fuse_bn_finalize and self.training and not sliced and nt_stats and g.stat_sync is None and self.affs is None
        if self.training and not sliced:
            stats = self.buf(f'bn.stats{i}', (2 * C + 1,), dtype=torch.float64)
            if fused:
                pass
            elif nt_stats:
                hipops.bn_reduce_partials(part, nt_stats, C, B * Lo, stats)
            else:
                hipops.bn_stats(self.xr, stats, self.buf('bn.partial', (2 * max(C, 256) * 64,), dtype=torch.float64))
            if g.stat_sync is not None:
                sync = g.stat_sync
                self.timed(f'stat_sync.{i}', self.S.py, lambda: sync(stats))
        a_t, s_t = self.buf(f'bn.a{i}', (B, C)), self.buf(f'bn.s{i}', (B, C))
        self.need('cond')              # (bf16 storage: gamma / beta - or the eval-mode affines - are the side stream's second step)
        if not self.st:                # (fp32: the side stream only carries gamma / beta - joined as late as their first use, which
            self.join_side()           # matters at B = 1, where conv_pre is shorter than the conditioning chain)
        if sliced:
            sl = self.buf(f'bn.slices{i}', (hipops.BN_SLICES * 2 * C,), dtype=torch.float64)
            hipops.bn_reduce_finalize_slices(part, nt_stats, B * Lo, sl, self.gbs[i], bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                             a_t, s_t, momentum=bn.momentum, eps=bn.eps)
        elif fused:
            hipops.bn_reduce_finalize(part, nt_stats, B * Lo, stats, self.gbs[i], bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                      a_t, s_t, momentum=bn.momentum, eps=bn.eps)
        elif self.affs is None:
            hipops.bn_finalize(stats, self.gbs[i], bn.running_mean, bn.running_var, bn.num_batches_tracked, a_t, s_t,
                               training=self.training, momentum=bn.momentum, eps=bn.eps)
        return a_t, s_t

    # ---------------------------------------------------------------------------------------------------------------------------
    def residual(self, i):
        """K6 / K7: the num_kernels residual blocks read the same x = a * xr + s; their mean is the next input (models.py:133-141)."""
        from .models import ResBlock2
        g, nk = self.g, self.nk
        self.rbs = [g.resblocks[i * nk + j] for j in range(nk)]
        self.names = [f'resblocks.{i * nk + j}' for j in range(nk)]
        # a back-propagated forward in the bf16 arithmetic (fp32 tensors): the 32-channel stage's convs layer by layer on the bf16 kernel too
        # (32 x 256 tiles; fragments packed here from the folded weights - the arena holds this stage's in the fused kernel's unit order)
        self.wps32 = {}
        if (self.save is not None and g.precision == 'bf16' and not self.st and self.C == 32 and self.Lo % 4 == 0 and self.algo == hipops.ALGO_AUTO
                and all(isinstance(rb, ResBlock2) and rb.kernel_size >= 3 and rb.kernel_size % 2 == 1 for rb in self.rbs)):
            for nm in self.names:
                for c in (0, 1):
                    self.wps32[f'{nm}.convs.{c}'] = hipops.pack_split(self.wf[f'{nm}.convs.{c}'], bf16=True)
        if not (self.algo != hipops.ALGO_DIRECT and nk <= 3):
            return self.residual_serial(i)
        B, C, Lo = self.B, self.C, self.Lo
        # The branches are independent until the final sum: conv n of ALL branches goes out as one launch (heaviest kernel size first); the
        # first nk - 1 branches end in their own buffers o_j and the last branch's final conv adds them in the reference's order
        # ((r0 + r1) + r2) / nk.
        self.t1s = [self.buf(f'act.t1_{i}_{j}', (B, C, Lo), dtype=self.adt) for j in range(nk)]
        self.outs = [self.buf(f'act.o_{i}_{j}', (B, C, Lo), dtype=self.adt) for j in range(nk - 1)] + [self.xs]
        self.heavy_first = sorted(range(nk), key=lambda j: -self.rbs[j].kernel_size)
        if not isinstance(self.rbs[0], ResBlock2):
            return self.residual1(i)
        self.all_wps = all(f'{nm}.convs.{c}' in self.wps for nm in self.names for c in (0, 1))
        self.all_wp = all(self.wp.get(f'{nm}.convs.{c}') is not None for nm in self.names for c in (0, 1))
        for cand in (self.rb2_stage_with_upsampler, self.rb2_narrow_stage_split, self.rb2_wide_stage_bf16, self.rb2_stage_f32,
                     self.rb2_stage_small, self.rb2_pairs, self.rb2_branch_convs_bf16, self.rb2_layers):
            if cand(i):
                return
            if self.st and cand is self.rb2_wide_stage_bf16 and C in (16, 32):
                raise RuntimeError('bf16 storage: the fused narrow-stage kernel did not take this shape (set generator.bf16_storage = False)')

    def launch(self, tag_sfx, probs):
        probs = [probs[j] for j in self.heavy_first if j in probs]
        tag = '+'.join(f'{self.names[j]}.{tag_sfx}' for j, _ in probs)
        self.timed(tag, hipops.conv1d_multi, [pr for _, pr in probs], splitk_ws=self.slab)

    def final_kw(self, j):
        if j < self.nk - 1:
            return {}
        return dict(add=self.outs[:self.nk - 1], out_div=float(self.nk))

    def stage_tag(self, sfx=''):
        return 'stage:' + '+'.join(f'{nm}.0&1' for nm in self.names) + sfx

    def rb2_stage_with_upsampler(self, i):
        """The stage with the NEXT stage's upsampler behind it in one kernel (bf16 tensors): xs is never written, the kernel stores
        act.up{i+1} and the BatchNorm partial sums of stage i + 1."""
        g, C, B, Lo, rbs = self.g, self.C, self.B, self.Lo, self.rbs
        if not (self.st and g.fuse_up and i + 1 < self.ns and C >= 32 and (C >= 64 and g.fuse_wide_stage or C in self.fuse_stage)
                and f'ups.{i + 1}' in self.wps and self.all_wps):
            return False
        nup = g.ups[i + 1]
        if not (nup.kernel_size == 2 * nup.stride and nup.stride in (2, 4) and nup.out_channels * 2 == C):
            return False
        ntn = hipops.resblock2_stage_up_tiles(B, C, Lo, [rb.kernel_size for rb in rbs], [rb.convs[0].dilation for rb in rbs],
                                              [rb.convs[1].dilation for rb in rbs], slope=LRELU_SLOPE,
                                              up_k=nup.kernel_size, up_u=nup.stride, up_slope=LRELU_SLOPE)
        if not ntn:
            return False
        xr_n = self.buf(f'act.up{i + 1}', (B, nup.out_channels, Lo * nup.stride), dtype=self.adt)
        part_n = self.buf(f'bn.part{i + 1}', (ntn * nup.out_channels * 2,)) if self.training else None
        ok = self.timed(self.stage_tag(f'+ups.{i + 1}'), hipops.resblock2_stage_split, self.xr, self.aff, _branches(rbs, self.names, self.wps), None,
                        slope=LRELU_SLOPE, out_div=float(self.nk), bf16=True, io_bf16=3,
                        up=(self.wps[f'ups.{i + 1}'], nup.bias.detach(), xr_n, part_n, nup.kernel_size, nup.stride, LRELU_SLOPE))
        if ok:
            self.up_done = ntn
        return ok

    def rb2_narrow_stage_split(self, i):
        """C = 32 / 16 on split / bf16 fragments: the whole residual section as one kernel - on bf16 tensors the last (16-channel) stage with
        the generator's tail behind it (models.py:143-145): the stage's output never leaves the chip, y is written instead."""
        g, C, B, Lo, rbs = self.g, self.C, self.B, self.Lo, self.rbs
        if not (C in (16, 32) and C in self.fuse_stage and self.all_wps):
            return False
        branches = _branches(rbs, self.names, self.wps)
        # (the reference's block set and 7-tap tail: the weights-in-registers kernel takes them; any other set / k <= 9 tail would run on the
        # resident-tile template, measured slower than the two kernels: only with fuse_post = 'any')
        std_set = [(rb.kernel_size, rb.convs[0].dilation, rb.convs[1].dilation) for rb in rbs] == [(3, 1, 3), (7, 1, 3), (11, 1, 3)]
        kp = g.conv_post.kernel_size
        if self.st and g.fuse_post and i == self.ns - 1 and C == 16 and kp <= 9 and ((std_set and kp == 7) or g.fuse_post == 'any'):
            y = torch.empty((B, 1, Lo), device=self.dev, dtype=torch.float32)
            if self.timed(self.stage_tag('+conv_post'), hipops.resblock2_stage_split, self.xr, self.aff, branches, None, slope=LRELU_SLOPE,
                          out_div=float(self.nk), bf16=True, io_bf16=3, post=(self.wf['conv_post'], g.conv_post.bias.detach(), y, kp, 0.01)):
                self.y = y
                return True
        return self.timed(self.stage_tag(), hipops.resblock2_stage_split, self.xr, self.aff, branches, self.xs, slope=LRELU_SLOPE,
                          out_div=float(self.nk), bf16=g.precision == 'bf16', io_bf16=3 if self.st else 0)

    def rb2_wide_stage_bf16(self, i):
        """A wide stage on bf16 tensors: the WHOLE residual section in one kernel (v2w_stage_bf16_wide.hip): x read once, t1_j on chip, one fp32
        accumulator over the branches."""
        g = self.g
        if not (self.st and g.fuse_wide_stage and self.C >= 64 and self.all_wps):
            return False
        return self.timed(self.stage_tag(), hipops.resblock2_stage_split, self.xr, self.aff, _branches(self.rbs, self.names, self.wps), self.xs,
                          slope=LRELU_SLOPE, out_div=float(self.nk), bf16=True, io_bf16=3)

    def _f32_branches(self, w, k1, k2):
        return [dict(**{k1: w[nm + '.convs.0'], k2: w[nm + '.convs.1']}, b1=rb.convs[0].bias.detach(), b2=rb.convs[1].bias.detach(),
                     k=rb.kernel_size, dil1=rb.convs[0].dilation, dil2=rb.convs[1].dilation) for nm, rb in zip(self.names, self.rbs)]

    def rb2_stage_f32(self, i):
        """The whole residual section of a narrow stage in ONE f32-MFMA kernel: x read once, t1_j in LDS, sum in registers - the last
        (16-channel) stage with the generator's tail behind it (models.py:143-145; round 5): its output is neither written nor read back."""
        g = self.g
        if self.st or not (self.C in self.fuse_stage and self.all_wp):
            return False
        branches = self._f32_branches(self.wp, 'wp1', 'wp2')
        kp = g.conv_post.kernel_size
        if g.fuse_post and i == self.ns - 1 and self.C == 16 and kp <= 9 and kp % 2 == 1 and g.conv_post.in_channels == 16:
            y = torch.empty((self.B, 1, self.Lo), device=self.dev, dtype=torch.float32)
            if self.timed(self.stage_tag('+conv_post'), hipops.resblock2_stage, self.xr, self.aff, branches, None, slope=LRELU_SLOPE,
                          out_div=float(self.nk), post=(self.wf['conv_post'], g.conv_post.bias.detach(), y, kp, 0.01)):
                self.y = y
                return True
        return self.timed(self.stage_tag(), hipops.resblock2_stage, self.xr, self.aff, branches, self.xs, slope=LRELU_SLOPE, out_div=float(self.nk))

    def rb2_stage_small(self, i):
        """8 channels (the sixth stage of a x640 generator): below every MFMA tile - the whole section as one FMA kernel."""
        if not (self.C == 8 and 8 in self.fuse_stage and not self.st and self.nk <= 4
                and all(self.wf.get(f'{nm}.convs.{c}') is not None for nm in self.names for c in (0, 1))):
            return False
        return self.timed(self.stage_tag(), hipops.resblock2_stage_small, self.xr, self.aff, self._f32_branches(self.wf, 'wf1', 'wf2'), self.xs,
                          slope=LRELU_SLOPE, out_div=float(self.nk))

    def launch_pairs(self, tag_sfx, probs):
        """probs: {j: dict}; the first nk - 1 branches in one launch, the summing branch after them."""
        nk, done = self.nk, True
        for js in ([j for j in self.heavy_first if j in probs and j < nk - 1], [nk - 1] if nk - 1 in probs else []):
            if js and done:
                done = self.timed('+'.join(f'{self.names[j]}.{tag_sfx}' for j in js), hipops.resblock_pair_multi, [probs[j] for j in js])
        return done

    def fused_pair_ok(self, keys):
        return (not self.st and self.C in self.fuse_pairs and self.C in (16, 32)
                and all(self.wp.get(f'{nm}.{c}') is not None for nm in self.names for c in keys))

    def rb2_pairs(self, i):
        """Narrow stages (C = 32 / 16): both convs of a pair in ONE kernel, the intermediate stays in LDS."""
        if not self.fused_pair_ok(('convs.0', 'convs.1')):
            return False
        rbs, names, wp = self.rbs, self.names, self.wp
        return self.launch_pairs('0&1', {j: dict(x=self.xr, in_affine=self.aff, wp1=wp[names[j] + '.convs.0'], b1=rbs[j].convs[0].bias.detach(),
                                                 wp2=wp[names[j] + '.convs.1'], b2=rbs[j].convs[1].bias.detach(), out=self.outs[j],
                                                 k=rbs[j].kernel_size, dil1=rbs[j].convs[0].dilation, dil2=rbs[j].convs[1].dilation,
                                                 res_mode=0, slope=LRELU_SLOPE, **self.final_kw(j)) for j in range(self.nk)})

    def conv2_problems(self):
        rbs, names = self.rbs, self.names
        return {j: (j, (self.t1s[j], self.wf[names[j] + '.convs.1'], rbs[j].convs[1].bias.detach(), self.outs[j],
                        dict(k=rbs[j].kernel_size, dil=rbs[j].convs[1].dilation, slope=LRELU_SLOPE, res=self.t1s[j],
                             **self.ck(names[j] + '.convs.1'), **self.final_kw(j)))) for j in range(self.nk)}

    def launch_conv2(self, sfx, conv2):
        nk = self.nk
        if nk > 1:
            self.launch(sfx, {j: conv2[j] for j in range(nk - 1)})
        self.launch(sfx, {nk - 1: conv2[nk - 1]})

    def rb2_branch_convs_bf16(self, i):
        """A wide stage on bf16 tensors the one-kernel form declined: the first convs of all branches in ONE launch (x staged once), then the
        second convs in one launch on one accumulator (v2w_branch_convs_bf16_fwd; o_j never written)."""
        g, rbs, names = self.g, self.rbs, self.names
        if not (self.st and g.fuse_wide and self.C >= 64 and self.all_wps):
            return False
        ks = [rb.kernel_size for rb in rbs]
        if not self.timed('bconv0:' + '+'.join(f'{nm}.0' for nm in names), hipops.branch_convs_bf16, 0, [self.xr], self.aff,
                          [self.wps[nm + '.convs.0'][0] for nm in names], [rb.convs[0].bias.detach() for rb in rbs], self.t1s,
                          ks, [rb.convs[0].dilation for rb in rbs], slope=LRELU_SLOPE):
            return False
        if not self.timed('bconv1:' + '+'.join(f'{nm}.1' for nm in names), hipops.branch_convs_bf16, 1, self.t1s, None,
                          [self.wps[nm + '.convs.1'][0] for nm in names], [rb.convs[1].bias.detach() for rb in rbs], [self.xs],
                          ks, [rb.convs[1].dilation for rb in rbs], slope=LRELU_SLOPE, out_div=float(self.nk)):
            self.launch_conv2('1', self.conv2_problems())
        return True

    def rb2_layers(self, i):
        """Layer by layer: conv1 of the three branches in ONE launch, conv2 of branches 0 .. nk - 2 in one, the last branch's conv2 adds them."""
        rbs, names = self.rbs, self.names
        self.launch('0', {j: (j, (self.xr, self.wf[names[j] + '.convs.0'], rbs[j].convs[0].bias.detach(), self.t1s[j],
                                  dict(k=rbs[j].kernel_size, dil=rbs[j].convs[0].dilation, slope=LRELU_SLOPE, in_affine=self.aff, res=self.xr,
                                       res_affine=self.aff, **self.ck(names[j] + '.convs.0')))) for j in range(self.nk)})
        self.launch_conv2('1', self.conv2_problems())
        return True

    # ---------------------------------------------------------------------------------------------------------------------------
    def residual1(self, i):
        """ResBlock1 (models.py:37-44): three (dilated conv, conv) pairs per branch, merged over the branches."""
        g, nk, B, C, Lo, rbs, names = self.g, self.nk, self.B, self.C, self.Lo, self.rbs, self.names
        xas = [self.buf(f'act.xa_{i}_{j}', (B, C, Lo), dtype=self.adt) for j in range(nk)]
        xbs = [self.buf(f'act.xb_{i}_{j}', (B, C, Lo), dtype=self.adt) for j in range(nk)]
        srcs, src_aff, t1s = [self.xr] * nk, self.aff, self.t1s
        for n in range(3):
            dsts = [xas, xbs, self.outs][n]
            if self.save is not None:   # backward needs every sub-block's conv1 output: one buffer per n
                t1s = [self.buf(f'act.t1_{i}_{j}_{n}', (B, C, Lo)) for j in range(nk)]
            last_kw = (lambda j: self.final_kw(j)) if n == 2 else (lambda j: {})
            if self.st and hipops.resblock1_pairs_ok(B, C, Lo, [rb.kernel_size for rb in rbs], [rb.convs1[n].dilation for rb in rbs], [1] * nk, slope=LRELU_SLOPE):
                # bf16 tensors: pair n of every branch as one resident-tile launch - the dilated conv's output stays in LDS, the pair's residual
                # joins the output - and the last pair of the last branch adds the other branches' results.  (A pair whose resident tiles do not
                # fit - 256 channels at dilation 5 - runs conv by conv on the chunked bf16 kernel below, still on bf16 tensors.)
                brs = [dict(wps1=self.wps[f'{names[j]}.convs1.{n}'], b1=rbs[j].convs1[n].bias.detach(), wps2=self.wps[f'{names[j]}.convs2.{n}'],
                            b2=rbs[j].convs2[n].bias.detach(), k=rbs[j].kernel_size, dil1=rbs[j].convs1[n].dilation, dil2=1) for j in range(nk)]
                for js in ([list(range(nk))] if n < 2 or nk == 1 else [list(range(nk - 1)), [nk - 1]]):
                    last = n == 2 and js[-1] == nk - 1
                    tag = 'rb1:' + '+'.join(f'{names[j]}.{2 * n}&{2 * n + 1}' for j in js)
                    if not self.timed(tag, hipops.resblock1_pairs_bf16, [srcs[j] for j in js], src_aff, [brs[j] for j in js], [dsts[j] for j in js],
                                      slope=LRELU_SLOPE, out_div=float(nk) if last else 0.0, add=self.outs[:nk - 1] if last and nk > 1 else None):
                        raise RuntimeError('bf16 storage: the ResBlock1 pair kernel declined a shape its query accepted')
                srcs, src_aff = dsts, None
                continue
            ok = False
            if self.fused_pair_ok((f'convs1.{n}', f'convs2.{n}')):
                ok = self.launch_pairs(f'{2 * n}&{2 * n + 1}',
                                       {j: dict(x=srcs[j], in_affine=src_aff, wp1=self.wp[f'{names[j]}.convs1.{n}'], b1=rbs[j].convs1[n].bias.detach(),
                                                wp2=self.wp[f'{names[j]}.convs2.{n}'], b2=rbs[j].convs2[n].bias.detach(), out=dsts[j],
                                                k=rbs[j].kernel_size, dil1=rbs[j].convs1[n].dilation, dil2=1, res_mode=1, slope=LRELU_SLOPE,
                                                **last_kw(j)) for j in range(nk)})
            if not ok:
                self.launch(str(2 * n), {j: (j, (srcs[j], self.wf[f'{names[j]}.convs1.{n}'], rbs[j].convs1[n].bias.detach(), t1s[j],
                                                 dict(k=rbs[j].kernel_size, dil=rbs[j].convs1[n].dilation, slope=LRELU_SLOPE, in_affine=src_aff,
                                                      **self.ck(f'{names[j]}.convs1.{n}')))) for j in range(nk)})
                conv2 = {j: (j, (t1s[j], self.wf[f'{names[j]}.convs2.{n}'], rbs[j].convs2[n].bias.detach(), dsts[j],
                                 dict(k=rbs[j].kernel_size, dil=1, slope=LRELU_SLOPE, res=srcs[j], res_affine=src_aff,
                                      **self.ck(f'{names[j]}.convs2.{n}'), **last_kw(j)))) for j in range(nk)}
                if n < 2:
                    self.launch(str(2 * n + 1), conv2)
                else:
                    self.launch_conv2('5', conv2)
            srcs, src_aff = dsts, None

    def residual_serial(self, i):
        """Branch after branch, conv after conv (the scalar cross-check kernels, or more than three branches): `xs` accumulates."""
        from .models import ResBlock2
        g, nk, B, C, Lo = self.g, self.nk, self.B, self.C, self.Lo
        t1 = self.buf(f'act.t1_{i}', (B, C, Lo))
        for j, (rb, name) in enumerate(zip(self.rbs, self.names)):
            k = rb.kernel_size
            last = dict(accumulate=(j > 0), out_div=(float(nk) if j == nk - 1 else 0.0))
            if isinstance(rb, ResBlock2):
                c1, c2 = rb.convs[0], rb.convs[1]
                self.timed(name + '.0', hipops.conv1d, self.xr, self.wf[name + '.convs.0'], c1.bias.detach(), t1, k=k, dil=c1.dilation,
                           slope=LRELU_SLOPE, in_affine=self.aff, res=self.xr, res_affine=self.aff, splitk_ws=self.slab, **self.ck(name + '.convs.0'))
                self.timed(name + '.1', hipops.conv1d, t1, self.wf[name + '.convs.1'], c2.bias.detach(), self.xs, k=k, dil=c2.dilation,
                           slope=LRELU_SLOPE, res=t1, splitk_ws=self.slab, **self.ck(name + '.convs.1'), **last)
                continue
            xa, xb = self.buf(f'act.xa_{i}', (B, C, Lo)), self.buf(f'act.xb_{i}', (B, C, Lo))
            src, src_aff, dsts = self.xr, self.aff, [xa, xb, self.xs]
            for n in range(3):
                c1, c2 = rb.convs1[n], rb.convs2[n]
                self.timed(f'{name}.{2 * n}', hipops.conv1d, src, self.wf[f'{name}.convs1.{n}'], c1.bias.detach(), t1, k=k, dil=c1.dilation,
                           slope=LRELU_SLOPE, in_affine=src_aff, splitk_ws=self.slab, **self.ck(f'{name}.convs1.{n}'))
                self.timed(f'{name}.{2 * n + 1}', hipops.conv1d, t1, self.wf[f'{name}.convs2.{n}'], c2.bias.detach(), dsts[n], k=k, dil=1,
                           slope=LRELU_SLOPE, res=src, res_affine=src_aff, splitk_ws=self.slab, **self.ck(f'{name}.convs2.{n}'),
                           **(last if n == 2 else {}))
                src, src_aff = dsts[n], None

    # ---------------------------------------------------------------------------------------------------------------------------
    def tail(self):
        """K8: leaky_relu(0.01) -> conv_post -> tanh (unless the last stage's kernel has already done it)."""
        g = self.g
        if self.st:
            # bf16 storage: every launch on a side stream has an event behind it (weights(), cond()); when the main stream has waited for all of
            # them - and every forked stream has one, so each has rejoined - there is nothing left to join (two waits on the main queue less)
            forked = range(1, 1 + self.cond_sid)
            if not (self.g.merge_waits and self.marked <= self.needed and all(any(s == sid for _n, s in self.order) for sid in forked)):
                self.S.join()    # (whatever step nobody asked for; the side streams have long finished)
        elif not self.joined:
            self.S.join()
        if self.y is None:
            self.y = torch.empty((self.B, 1, self.L), device=self.dev, dtype=torch.float32)
            self.timed('conv_post', hipops.conv_post_tanh, self.cur, self.wf['conv_post'], g.conv_post.bias.detach(), self.y, k=7, slope=0.01)
        return self.y

    def hand_over(self, y):
        """What backward.py reads: this forward's buffers, folded weights and the spectral-norm vectors AS THIS FORWARD LEFT THEM (the backward
        of sigma = u^T W v must not see a later forward's power-iteration step)."""
        g, save = self.g, self.save
        save['sn_uv'] = [(c.layer.weight_u.detach().clone(), c.layer.weight_v.detach().clone()) for c in g.cbns]
        save.update(ws=g._ws, wf=self.wf, wp=self.wp, wpd=g._fold_key.get('wpd', {}), vers=g._fold_key.get('vers'), gen=g._fold_key.get('gen'),
                    y=y, x=self.x, spk=self.spk, nz=self.nz, training=self.training, B=self.B, T=self.T)
        g._fold_key.pop('state', None)     # the cached fold pointed into the handed-over buffers
