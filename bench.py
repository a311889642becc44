#!/usr/bin/env python3
"""Vec2Wav generator forward benchmark on MI355X (BASELINE.json metric: audio samples/sec).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one `Generator.forward` (train-mode CondBN: batch statistics, spectral-norm power iteration,
weight-norm fold all inside the timed region) over one synthetic batch resident in HBM.
Workload = BASELINE.json configs[1]: B=32 per GPU, T=256 frames, 768-d latents, x320, fp32, ResBlock2 (default hparams).
N>1: one process per GPU, weak scaling (B=32 per rank), CondBN statistics all-reduced over RCCL each stage.
`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment) starts the N ranks itself through
torch.distributed.run - before anything touches a GPU - and exits non-zero when fewer than N devices are visible:
it never reports a smaller run than the one asked for.  Rank 0 prints ONE JSON line.
"""
import argparse
from statistics import mean, median
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_BF16_MFMA_TFLOPS = 2516.8  # MI355X_MICROARCH.md: dense bf16 MFMA = 16 x the f32 rate (~2.5 PF)


def baseline_metric() -> str:
    """The metric string of BASELINE.json (the file travels with the repo); the built-in name if it is missing."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'BASELINE.json')) as f:
            return json.load(f)['metric']
    except Exception:
        return 'audio samples/sec (16 kHz) Vec2Wav generator forward'


def usable_cpus() -> int:
    """CPUs this process may actually run on: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _timed_oracle(O, sd, h, inp, training, threads, reps, warmups=1):
    import torch
    torch.set_num_threads(threads)
    for _ in range(warmups):
        O.generator_forward(sd, h, *inp, training=training)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        O.generator_forward(sd, h, *inp, training=training)
        ts.append(time.perf_counter() - t0)
    return median(ts)


def cpu_baseline(h):
    """The oracle (CPU restatement of the reference forward, oracle/vec2wav_oracle.py) timed on this box's host cores: a reported
    baseline, never the target.  Bounded to about a minute:
      * cfg1 (BASELINE configs[0]: B=1, T=50, the reference's own CPU-runnable case): 1 thread and all usable cores, train and eval
        mode, median of 10 after 3 warm-ups (SURVEY 8(d) / BASELINE.md 4);
      * the cfg2 sample B=4 x T=256 (1/8 of the cfg2 batch, same per-sample work) in train mode: short sweep over thread counts,
        best median of 3 -> `cfg2_sample_B4` (more threads than the problem can feed are slower);
      * cfg2 ITSELF (B=32 x T=256) once at all usable cores after one warm-up -> `value` / `cores`."""
    from oracle import vec2wav_oracle as O
    from wavthruvec_pytorch_amd import synthetic
    sd = synthetic.make_state_dict(h, seed=0)
    ncpu = usable_cpus()
    up = synthetic.total_upsample(h)
    t_end = time.time() + 60.0
    cfg1 = {}
    inp1 = synthetic.make_inputs(h, 1, 50, seed=1234)
    for mode, training in (('train', True), ('eval', False)):
        for th in sorted({1, ncpu}):
            if time.time() > t_end:
                break
            cfg1[f'{mode}_threads{th}'] = 50 * up / _timed_oracle(O, sd, h, inp1, training, th, 10, warmups=3)
    B, T = 4, 256
    inp = synthetic.make_inputs(h, B, T, seed=1234)
    cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu} or {ncpu})
    best = None
    for th in cands:
        med = _timed_oracle(O, sd, h, inp, True, th, 3)
        if best is None or med < best[0]:
            best = (med, th)
        if time.time() > t_end:
            break
    med, th = best
    small = dict(value=B * T * up / med, unit='samples/s', cores=th,
                 sample=f'train mode, B={B} T={T} 768-d (1/8 of the cfg2 batch), median of 3 at {th} threads (best of thread counts {cands})')
    # `value` is the bench line's OWN workload: one full cfg2 forward (B=32, T=256, train mode) at all usable cores after one warm-up
    # (~5 s each); the B=4 sample - which a smaller thread count serves better - stays beside it
    inpf = synthetic.make_inputs(h, 32, 256, seed=1234)
    tf = _timed_oracle(O, sd, h, inpf, True, ncpu, 1, warmups=1)
    return dict(value=32 * 256 * up / tf, unit='samples/s', cores=ncpu, kind='port', seconds_per_forward=tf,
                sample=f'oracle (torch CPU fp32 restatement of the reference forward), the bench line\'s workload itself: ONE full cfg2 forward '
                       f'(B=32, T=256, 768-d, train mode) at {ncpu} threads after one warm-up ({ncpu} usable CPUs)',
                cfg2_sample_B4=small,
                cfg1_B1_T50_samples_per_s=cfg1, cfg1_protocol='median of 10 after 3 warm-ups (BASELINE.md 4)')


def self_launch(args) -> int:
    """`--gpus N` (N > 1) without a launcher: start N ranks with torch.distributed.run as a CHILD process and pass its exit
    code on.  Nothing here initialises the GPU (torch.cuda.device_count() does not), so no initialised process forks or execs."""
    import torch
    have = torch.cuda.device_count()
    pinned = os.environ.get('V2W_BENCH_DEVICE') is not None      # test hook: every rank on one device
    if not args.dry_run and have < (1 if pinned else args.gpus):
        print(f'bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible', file=sys.stderr)
        return 3
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


def csrc_hash() -> str:
    """Hash of the kernel sources and headers (wavthruvec_pytorch_amd.build.sources_hash): a PMC traffic file collected from other
    sources is stale."""
    from wavthruvec_pytorch_amd import build
    return build.sources_hash()


def traffic_of(kernel: str, suffix: str):
    """HBM bytes per launch of `kernel` from the latest committed PMC pass profiles/rNN_<suffix> (tools/pmc_traffic.py: FETCH_SIZE x2 +
    WRITE_SIZE, KiB -> bytes as MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside this process.  The file records a
    hash of the kernel sources it was collected from; when csrc/ has changed since, the number is withheld (null + the reason)."""
    pdir = os.path.join(ROOT, 'profiles')
    tpaths = sorted(q for q in os.listdir(pdir) if q.endswith(suffix)) if os.path.isdir(pdir) else []
    if not tpaths:
        return None, None
    with open(os.path.join(pdir, tpaths[-1])) as f:
        tj = json.load(f)
    meta = tj.get('_meta', {})
    want = kernel.replace(' ', '')
    hit = sorted((k for k in tj if k != '_meta' and k.replace(' ', '').startswith(want)), key=lambda k: -tj[k].get('launches', 0))
    if not hit:
        return None, f'profiles/{tpaths[-1]}: no entry for {kernel}'
    if meta.get('csrc_sha') != csrc_hash():
        return None, (f'stale: csrc changed since profiles/{tpaths[-1]} was collected (commit {meta.get("commit", "?")}, '
                      f'sources {meta.get("csrc_sha", "unrecorded")}, now {csrc_hash()})')
    return tj[hit[0]]['hbm_bytes_per_launch'], (f'profiles/{tpaths[-1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; FETCH x2 '
                                                 f'per the gfx950 correction; collected {meta.get("date", "?")} at commit {meta.get("commit", "?")}, '
                                                 f'same kernel sources)')


def schedule_traffic(per_kernel: dict, suffix: str, step_s: float):
    """The roofline of the schedule that RAN, beside the layer-granular contract figure of SURVEY 8(d): the fused kernels never move most of
    the per-layer bytes (a stage kernel reads its input once and writes the next stage's input), so `hbm_frac` on the algorithmic count says
    how far the forward is from the 8(d) target, not how busy HBM is.  Here: counter bytes (profiles/rNN_<suffix>: FETCH_SIZE x 2 + WRITE_SIZE
    per launch, collected from the same kernel sources) x launches per step, summed over the step's conv kernels, / step time / 8 TB/s.
    None (+ the reason) when a kernel of the step has no row in the profile or the profile is stale - a name the library reports and the
    cited rocprof summary does not hold is an error of the evidence, and is said so."""
    if not suffix:
        return None
    total, missing, src = 0.0, [], None
    for k, v in per_kernel.items():
        names = [n.strip() for n in k.split(' + ')]
        for n in names:
            t, why = traffic_of(n, suffix)
            if t is None:
                missing.append(f'{n}: {why}')
            else:
                total += t * v['launches']
                src = why
    if missing:
        return dict(counter_bytes_per_step=None, hbm_frac_counter=None, missing=missing)
    return dict(counter_bytes_per_step=total, hbm_frac_counter=total / step_s / 1e9 / PEAK_HBM_GBS, source=src)


def run_steps(g, inp, steps, warmup, barrier=lambda: None):
    """EXACTLY `steps` forwards between barrier + synchronize pairs (wall clock), HIP events on the launching stream around every
    step as well (event-timed median).  Returns (elapsed seconds, per-step milliseconds)."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    with torch.no_grad():
        for _ in range(warmup):
            g(*inp)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            g(*inp)
            evs[i + 1].record()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
    return elapsed, [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]


def roofline_block(g, h, inp, B, T, precision, algo, step_s, traffic_suffix, record=True):
    """Roofline of the dominant kernel of `g`'s forward: HIP events around every conv launch on the launching stream (3 profiled
    forwards, run by every rank - they contain the statistics all-reduce; `record` = this rank keeps the numbers), algorithmic
    FLOPs / bytes of the layers each launch covers (workmodel.py = SURVEY 8(d)), grouped by the kernel instantiation that ran them."""
    from wavthruvec_pytorch_amd import workmodel, hipops
    per, sync_t = {}, []
    with torch.no_grad():
        for _ in range(3):
            g._profile = [] if record else None
            g(*inp)
            torch.cuda.synchronize()
            if record:
                for tag, e0, e1 in g._profile:
                    (sync_t if tag.startswith('stat_sync.') else per.setdefault(tag, [])).append(e0.elapsed_time(e1) * 1e-3)
        g._profile = None
    # which kernel(s) ran each tagged launch: asked of the LIBRARY (v2w_name_sink, ABI v33: every launching entry point reports the kernels it
    # selects, as rocprofv3 prints them) - this file holds no kernel name and no tile table.  One more forward, on every rank (a data-parallel
    # forward contains the statistics all-reduce).
    tag_kernels = g.profile_kernel_names(*inp)
    if not record:
        return None
    bf16_run = precision == 'bf16'
    act_bytes = 2 if (bf16_run and g.bf16_storage) else 4      # bytes of an activation element between layers in this mode
    peak_tf = PEAK_BF16_MFMA_TFLOPS if bf16_run else PEAK_FP32_MFMA_TFLOPS
    layers = {l['name']: l for l in workmodel.conv_layers(h, B, T, act_bytes)}
    launches = {}      # tag -> dict(kernel, flops, bytes, t)
    for tag, ts in per.items():
        names, fused = [], False
        staged = tag.startswith('stage:')       # whole residual section of a narrow stage in one kernel
        for part in tag.replace('stage:', '').split(':')[-1].split('+'):   # 'resblocks.J.a&b' = convs a and b of block J fused
            if '&' in part:
                base, ab = part.rsplit('.', 1)
                names += [f'{base}.{x}' for x in ab.split('&')]
                fused = True
            else:
                names.append(part)
        ls = [layers[n] for n in names]
        bconv = tag.startswith('bconv')           # v2w_branch_convs_bf16_fwd: the first / second convs of a wide stage's branches in one launch
        if bconv:
            tag_names = tag.split(':', 1)[1]
            names = tag_names.split('+')
            ls = [layers[n] for n in names]
        ran = list(dict.fromkeys(tag_kernels.get(tag, [])))      # distinct kernels behind this tag, in launch order
        if not ran:
            raise RuntimeError(f'bench: the library reported no kernel for the launch tagged {tag!r}')
        kname = ran[0] if len(ran) == 1 else ' + '.join(ran)
        nbytes = sum(l['bytes'] for l in ls)
        if fused:                              # the intermediate is neither written nor re-read
            nbytes -= sum(2 * B * l['cout'] * l['L'] * act_bytes for l in ls[::2])
        if staged:                             # x read once, no per-branch outputs / running-sum traffic
            act = B * ls[0]['cout'] * ls[0]['L'] * act_bytes
            nbytes = 2 * act + sum(l['cin'] * l['cout'] * l['k'] * 4 for l in ls)
            if names[-1] == 'conv_post':          # the fused tail: the stage's output is not written, the fp32 audio is
                nbytes += B * ls[0]['L'] * 4 - act
            if names[-1].startswith('ups.'):      # the fused upsampler: the stage's output is not written, the next stage's input is
                lu = ls[-1]
                nbytes = act + B * lu['cout'] * lu['L'] * lu['u'] * act_bytes + sum(l['cin'] * l['cout'] * l['k'] * 4 for l in ls)
        launches[tag] = dict(kernel=kname, flops=sum(l['flops'] for l in ls), bytes=nbytes,
                             t=mean(ts), conv=all(l['name'] != 'conv_post' for l in ls))
    groups = {}
    for tag, d in launches.items():
        groups.setdefault(d['kernel'], []).append(tag)
    gtime = {k: sum(launches[n]['t'] for n in v) for k, v in groups.items()}
    gflops = {k: sum(launches[n]['flops'] for n in v) for k, v in groups.items()}
    dom = max(gtime, key=gtime.get)
    dom_tags = groups[dom]
    dom_t, dom_f = gtime[dom], gflops[dom]
    conv_t = sum(d['t'] for d in launches.values() if d['conv'])
    conv_f = sum(d['flops'] for d in launches.values() if d['conv'])
    all_t = sum(d['t'] for d in launches.values())
    tot_f, tot_b = workmodel.totals(h, B, T, act_bytes)
    traffic, traffic_src = traffic_of(dom, traffic_suffix) if traffic_suffix else (None, None)
    dom_b = sum(launches[n]['bytes'] for n in dom_tags)
    mfma_frac, hbm_frac = dom_f / dom_t / 1e12 / peak_tf, dom_b / dom_t / 1e9 / PEAK_HBM_GBS
    bound = 'hbm' if hbm_frac > mfma_frac else 'mfma'        # the ceiling the dominant kernel sits closer to
    roof = dict(bound=bound, kernel=dom + (' (bf16 MFMA implicit-GEMM conv)' if bf16_run else ' (f32 MFMA implicit-GEMM conv)'),
                launches=dom_tags,
                achieved=(dom_b / dom_t / 1e9) if bound == 'hbm' else (dom_f / dom_t / 1e12),
                peak=PEAK_HBM_GBS if bound == 'hbm' else peak_tf, unit='GB/s' if bound == 'hbm' else 'TFLOP/s',
                frac=max(mfma_frac, hbm_frac), mfma_frac=mfma_frac, hbm_frac=hbm_frac,
                traffic=traffic, traffic_source=traffic_src,
                algorithmic_bytes_per_launch_avg=dom_b / len(dom_tags),
                launches_per_step=len(dom_tags), avg_launch_us=dom_t / len(dom_tags) * 1e6,
                flops_per_launch_avg=dom_f / len(dom_tags), activation_bytes_per_element=act_bytes,
                per_kernel={k: dict(launches=len(v), ms=round(gtime[k] * 1e3, 4), tflops=round(gflops[k] / gtime[k] / 1e12, 2),
                                    algorithmic_gbs=round(sum(launches[n]['bytes'] for n in v) / gtime[k] / 1e9, 1))
                            for k, v in sorted(groups.items(), key=lambda kv: -gtime[kv[0]])},
                all_conv_launches=dict(achieved=conv_f / conv_t / 1e12, frac=conv_f / conv_t / 1e12 / peak_tf,
                                       launches_per_step=sum(1 for d in launches.values() if d['conv'])),
                whole_forward=dict(flops=tot_f, algorithmic_bytes=tot_b, sum_conv_kernel_ms=all_t * 1e3,
                                   mfma_frac=tot_f / step_s / 1e12 / peak_tf,
                                   hbm_frac=tot_b / step_s / 1e9 / PEAK_HBM_GBS))
    if sync_t:
        roof['stat_sync_allreduce_us_mean'] = mean(sync_t) * 1e6
    roof['schedule'] = schedule_traffic(roof['per_kernel'], traffic_suffix, step_s)
    return roof


def single_rank_rccl_group(dev):
    """A ONE-rank RCCL communicator on this GPU (`init_process_group('nccl', world_size=1, device_id=...)`): the code path a multi-GPU
    job runs - communicator creation, the device all-reduce on the compute stream, the barrier - executed on a one-GPU box."""
    if dist.is_initialized():
        return
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)


class stdout_to_stderr:
    """RCCL prints its version banner to the C-level stdout when a communicator is created; rank 0's stdout carries ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.keep = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:                                   # the banner sits in libc's stdout buffer (a file or pipe is block-buffered): out with it
            import ctypes                      # while fd 1 still is stderr
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.keep, 1)
        os.close(self.keep)


def stat_sync_overhead(g, inp, dev, steps):
    """What the data-parallel CondBN exchange costs per forward, measured on the hardware at hand: the same train-mode step with and
    without `enable_sync_batchnorm()` over a one-rank RCCL group (five sequentially dependent fp64 all-reduces of [sum | sumsq | count]
    on the compute stream), interleaved rounds in one process, plus the event-timed duration of each all-reduce.  At one
    rank the collective moves no data: this is its fixed launch + completion latency, the floor of the per-stage cost at N ranks."""
    single_rank_rccl_group(dev)
    keep = g.stat_sync
    g.stat_sync = None
    with_ms, without_ms = [], []
    from wavthruvec_pytorch_amd.distributed import BNStatSync
    sync = BNStatSync(single_rank_collective=True)     # (a one-rank group skips the exchange unless asked: here it is what is timed)
    for _ in range(3):
        for on in (True, False):
            g.stat_sync = sync if on else None
            el, _ms = run_steps(g, inp, steps, 2)
            (with_ms if on else without_ms).append(el / steps * 1e3)
    g.stat_sync = sync
    per = []
    with torch.no_grad():
        for _ in range(3):
            g._profile = []
            g(*inp)
            torch.cuda.synchronize()
            per += [e0.elapsed_time(e1) * 1e3 for tag, e0, e1 in g._profile if tag.startswith('stat_sync.')]
        g._profile = None
    g.stat_sync = keep
    w, wo = median(with_ms), median(without_ms)
    return dict(backend=sync.backend, ranks=sync.world_size, allreduces_per_forward=g.num_upsamples,
                ms_per_step_with_sync=w, ms_per_step_without_sync=wo, stat_sync_ms_per_step=w - wo,
                allreduce_us_event_mean=mean(per) if per else None, allreduce_us_event_max=max(per) if per else None,
                rounds_with=with_ms, rounds_without=without_ms,
                note='one-rank RCCL group on one GPU: the fixed cost of the five per-stage all-reduces (no bytes cross a link); '
                     'at N ranks each adds the xGMI all-reduce latency of <= 4 KiB')


def config_block(hp, B, T, precision, steps, warmup, dev, workload, parity, traffic_suffix=None):
    """One more single-GPU BASELINE configuration as a first-class block of the bench line: its own generator, its own timed steps
    (wall clock + event median), priced against both ceilings with ITS byte count, and the roofline of its dominant kernel."""
    from wavthruvec_pytorch_amd import Generator, synthetic, workmodel
    h = synthetic.make_hparams(**hp)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).train()
    g.precision = precision
    inp = synthetic.make_inputs(h, B, T, seed=4, device=dev)
    el, ms = run_steps(g, inp, steps, warmup)
    act = 2 if precision == 'bf16' else 4
    fl, by = workmodel.totals(h, B, T, act)
    st = el / steps
    peak = PEAK_BF16_MFMA_TFLOPS if precision == 'bf16' else PEAK_FP32_MFMA_TFLOPS
    roof = roofline_block(g, h, inp, B, T, precision, 'auto', st, traffic_suffix)
    up = synthetic.total_upsample(h)
    del g
    torch.cuda.empty_cache()
    return dict(workload=workload, dtype=precision, steps=steps, ms_per_step=st * 1e3, ms_per_step_event_median=median(ms),
                value=B * T * up / st, unit='samples/s', flops=fl, algorithmic_bytes=by,
                hbm_frac=by / st / 1e9 / PEAK_HBM_GBS, mfma_frac=fl / st / 1e12 / peak,
                peaks=dict(hbm_gbs=PEAK_HBM_GBS, mfma_tflops=peak), roofline=roof, parity=parity)


def inference_block(dev, steps, warmup):
    """The forward the inference pipeline runs (synthesize.py; reference inference.py): EVAL mode - running statistics, no spectral-norm step,
    weights folded once and kept - in the configs[2] arithmetic at the north_star's shape and at configs[2]'s.  Time, throughput and both
    fractions per shape; the per-kernel rooflines are the train-mode blocks' (same conv kernels, without the statistics launches between them)."""
    from wavthruvec_pytorch_amd import Generator, synthetic, workmodel
    h = synthetic.make_hparams(num_wv_feat=768)
    up = synthetic.total_upsample(h)
    out = dict(workload='Generator.forward in EVAL mode (running statistics, cached weight fold: the inference pipeline), 768-d latents, x320, '
                        'bf16 compute / fp32 accumulate, bf16 activation storage', dtype='bf16', steps=steps,
               parity='tests/test_hip_generator.py::test_generator_cfg2_bf16_full_size_vs_oracle_eval (the reference autocast bar, eval mode)')
    for name, B, T in (('cfg2', 32, 256), ('cfg3', 64, 512)):
        g = Generator(h)
        g.load_state_dict(synthetic.make_state_dict(h, seed=0))
        g = g.to(dev).eval()
        g.precision = 'bf16'
        inp = synthetic.make_inputs(h, B, T, seed=4, device=dev)
        el, ms = run_steps(g, inp, steps, warmup)
        fl, by = workmodel.totals(h, B, T, 2)
        st = el / steps
        out[name] = dict(B=B, T=T, ms_per_step=st * 1e3, ms_per_step_event_median=median(ms), value=B * T * up / st, unit='samples/s',
                         hbm_frac=by / st / 1e9 / PEAK_HBM_GBS, mfma_frac=fl / st / 1e12 / PEAK_BF16_MFMA_TFLOPS)
        del g
        torch.cuda.empty_cache()
    # the block's own headline = the north_star shape
    out.update({k: out['cfg2'][k] for k in ('ms_per_step', 'ms_per_step_event_median', 'value', 'unit', 'hbm_frac', 'mfma_frac')})
    return out


def train_step_block(dev, B, T, steps, precision='f32'):
    """The generator half of a vec2wav/train.py:204-215 step at the cfg2 shape: forward (autograd schedule) + backward through the C ABI +
    AdamW (train.py:100 betas / lr), a weighted-sum loss.  FLOPs = 3 x the forward's (input- and weight-gradient GEMMs).  precision 'f32':
    exact fp32 everywhere; 'f16x3': the wide convs of the forward and their input gradients on the f16 matrix pipe with split operands
    (~22-bit products, fp32 accumulate), weight gradients exact fp32."""
    from wavthruvec_pytorch_amd import Generator, synthetic, workmodel
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).train()
    g.precision = precision
    opt = torch.optim.AdamW(g.parameters(), 2e-4, betas=(0.8, 0.99))
    inp = synthetic.make_inputs(h, B, T, seed=1, device=dev)
    up = synthetic.total_upsample(h)
    dy = torch.randn(B, 1, T * up, device=dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for it in range(steps + 2):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            evs[0].record()
        opt.zero_grad(set_to_none=True)
        (g(*inp) * dy).sum().backward()
        opt.step()
        if it >= 2:
            evs[it - 1].record()
    torch.cuda.synchronize()
    st = (time.perf_counter() - t0) / steps
    fl, _by = workmodel.totals(h, B, T, 4)
    del g, opt
    torch.cuda.empty_cache()
    return dict(workload=f'generator training step (vec2wav/train.py:204-215 without the discriminators): forward + backward + AdamW, B={B} x T={T}, '
                         '768-d, x320, ResBlock2, ' + {'f32': 'exact fp32', 'f16x3': 'forward and input-gradient convs as f16 hi+lo (3 MFMA per product, fp32 '
                                                      'accumulate), weight gradients exact fp32',
                                                      'bf16': 'the reference autocast arithmetic on fp32 tensors: forward, input- and weight-gradient '
                                                      'convs from bf16 operands with fp32 accumulation (v2w_wgrad_bf16) wherever the layer has '
                                                      'that kernel'}[precision],
                dtype=precision, steps=steps, ms_per_step=st * 1e3,
                ms_per_step_event_median=median(evs[i].elapsed_time(evs[i + 1]) for i in range(steps)),
                value=B * T * up / st, unit='trained samples/s', flops=3.0 * fl, tflops=3.0 * fl / st / 1e12,
                mfma_frac=3.0 * fl / st / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                parity='tests/test_hip_generator.py::test_generator_backward_matches_oracle_autograd (every parameter gradient vs autograd through the oracle)')


def dry_run(args, rank, world) -> int:
    """`--dry-run`: everything of an N-rank run except the GPU work - process group over V2W_BENCH_BACKEND (gloo on a CPU box), the
    per-rank input seeds and batch shards, barrier, max-over-ranks, ONE JSON line from rank 0 with `value` null.  What an 8-GPU node
    meets first (rendezvous of 8 processes, rank -> seed / shard bookkeeping, stdout discipline) is checked without hardware."""
    backend = os.environ.get('V2W_BENCH_BACKEND', 'nccl')
    if backend == 'nccl' and not torch.cuda.is_available():
        raise SystemExit('--dry-run without a GPU needs V2W_BENCH_BACKEND=gloo')
    redirect = stdout_to_stderr()
    redirect.__enter__()
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)
    from wavthruvec_pytorch_amd import synthetic
    from wavthruvec_pytorch_amd.distributed import shard_bounds
    h = synthetic.make_hparams(num_wv_feat=768)
    B, T = args.batch, args.frames
    seed = 1234 + rank
    x, spk, nz = synthetic.make_inputs(h, 1, 4, seed=seed)           # (a sliver of this rank's inputs: enough to tell the seeds apart)
    mine = dict(rank=rank, seed=seed, shard=list(shard_bounds(B * world, rank, world)), x_checksum=float(x.double().sum()))
    allr = [None] * world
    if world > 1:
        dist.barrier()
        dist.all_gather_object(allr, mine)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == world - 1
    else:
        allr = [mine]
    if dist.is_initialized():
        dist.destroy_process_group()
    redirect.__exit__()
    if rank == 0:
        up = synthetic.total_upsample(h)
        print(json.dumps({
            'metric': baseline_metric(), 'value': None, 'unit': 'samples/s', 'n_gpus': world, 'rccl_ranks': None, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic', 'dry_run': dict(backend=backend, ranks=allr, samples_per_step=world * B * T * up),
            'config': {'workload': f'DRY RUN (no GPU work) of BASELINE configs[1]: B={B}/GPU x T={T} frames', 'global_batch': B * world, 'frames': T,
                       'parallelism': f'dp{world} (batch shards, all-reduce of CondBN stats)' if world > 1 else 'single GPU'}}), flush=True)
    return 0


def _r(v, nd=4):
    """Round floats for the compact line (the full-precision values are in the detail file / --verbose)."""
    if isinstance(v, float):
        return float(f'{v:.{nd}g}') if abs(v) < 1 else round(v, nd)
    return v


def _compact_schedule(sc):
    """{counter_bytes_per_step, hbm_frac_counter} - or, when the cited profile is stale / lacks a kernel, how many kernels and the first reason."""
    out = {k: _r(sc.get(k)) for k in ('counter_bytes_per_step', 'hbm_frac_counter')}
    if sc.get('missing'):
        out['missing'] = [len(sc['missing']), str(sc['missing'][0])[:72]]
    return out


def compact_roofline(roof):
    """The keys the contract names (bound, achieved, peak, unit, frac, traffic) + what identifies the kernel; the per-kernel table and
    the launch tag lists stay in the detail file (`--verbose` puts them back into the line)."""
    if not isinstance(roof, dict):
        return roof
    keep = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'mfma_frac', 'hbm_frac', 'traffic', 'launches_per_step', 'avg_launch_us',
            'algorithmic_bytes_per_launch_avg', 'flops_per_launch_avg', 'stat_sync_allreduce_us_mean')
    out = {k: _r(roof[k]) for k in keep if k in roof}
    src = roof.get('traffic_source')
    if src:
        out['traffic_source'] = src if len(src) <= 96 else src[:93] + '...'
    wf = roof.get('whole_forward')
    if wf:
        out['whole_forward'] = {k: _r(wf[k]) for k in ('mfma_frac', 'hbm_frac', 'sum_conv_kernel_ms') if k in wf}
    sc = roof.get('schedule')
    if isinstance(sc, dict):
        out['schedule'] = _compact_schedule(sc)
    return out


def compact_block(b):
    """One extra block of the line: its step time, throughput, both roofline fractions and its dominant kernel's fraction."""
    if not isinstance(b, dict) or 'error' in b:
        return b
    keep = ('dtype', 'steps', 'ms_per_step', 'ms_per_step_event_median', 'value', 'unit', 'hbm_frac', 'mfma_frac', 'tflops',
            'ms_per_step_with_sync', 'ms_per_step_without_sync', 'stat_sync_ms_per_step', 'allreduce_us_event_mean', 'backend', 'ranks',
            'precision', 'max_abs_diff_vs_f32_path', 'parity_bar')
    out = {k: _r(b[k]) for k in keep if k in b}
    r = b.get('roofline')
    if isinstance(r, dict):
        out['dominant'] = {k: _r(r[k]) for k in ('kernel', 'bound', 'frac', 'avg_launch_us', 'traffic') if k in r}
        sc = r.get('schedule')
        if isinstance(sc, dict):      # the schedule that ran: counter bytes per step and their fraction of the HBM peak (None + why when the profile is stale)
            out['schedule'] = _compact_schedule(sc)
        if b.get('dtype') == 'bf16' and 'resblock' not in str(b.get('workload', '')).lower()[:40] and isinstance(r.get('per_kernel'), dict):
            # the bf16 pipeline's launches, largest first: [kernel, ms per step, TFLOP/s]
            out['kernels'] = [[k.replace(' ', ''), _r(v['ms']), _r(v['tflops'])] for k, v in list(r['per_kernel'].items())[:7]]
    for key in ('cfg2', 'cfg3'):          # inference_bf16: [B, T, ms per forward, hbm_frac, mfma_frac] per shape
        e = b.get(key)
        if isinstance(e, dict) and 'ms_per_step' in e:
            out[key] = [e.get('B'), e.get('T'), _r(e['ms_per_step']), _r(e.get('hbm_frac')), _r(e.get('mfma_frac'))]
    for key in ('alt_precision_f16x3', 'alt_precision_bf16'):
        alt = b.get(key)
        if isinstance(alt, dict):
            out[key] = {k: _r(alt[k]) for k in ('ms_per_step', 'value', 'tflops', 'error') if k in alt}
    return out


def compact_cpu(c):
    if not isinstance(c, dict):
        return c
    out = {k: _r(c[k]) for k in ('value', 'unit', 'cores', 'kind', 'seconds_per_forward') if k in c}
    out['sample'] = f"oracle (torch CPU fp32 restatement of the reference forward): ONE full cfg2 forward (B=32, T=256, train mode) at {c.get('cores')} threads after one warm-up"
    sm = c.get('cfg2_sample_B4')
    if isinstance(sm, dict):
        out['cfg2_sample_B4'] = {k: _r(sm[k]) for k in ('value', 'cores') if k in sm}
    c1 = c.get('cfg1_B1_T50_samples_per_s')
    if isinstance(c1, dict):
        out['cfg1_B1_T50_samples_per_s'] = {k: _r(v) for k, v in c1.items()}
    return out


def summary_of(out, blocks):
    """{block: [ms_per_step, hbm_frac (SURVEY 8(d) byte count), mfma_frac(, hbm_frac of the counter bytes the schedule really moved)]} - the LAST
    key of the line, so that it survives a record that keeps only the tail."""
    sm = {}
    wf = (out.get('roofline') or {}).get('whole_forward') or {}
    sm['cfg2_f32'] = [_r(out.get('ms_per_step')), _r(wf.get('hbm_frac')), _r(wf.get('mfma_frac'))]
    sc0 = (out.get('roofline') or {}).get('schedule') or {}
    if sc0.get('hbm_frac_counter') is not None:
        sm['cfg2_f32'].append(_r(sc0['hbm_frac_counter']))
    for name in blocks:
        b = out.get(name)
        if not isinstance(b, dict):
            continue
        if 'error' in b:
            sm[name] = 'error'
        elif name == 'stat_sync':
            sm[name] = [_r(b.get('stat_sync_ms_per_step')), _r(b.get('allreduce_us_event_mean')), None]
        else:
            sm[name] = [_r(b.get('ms_per_step')), _r(b.get('hbm_frac')), _r(b.get('mfma_frac'))]
            sc = ((b.get('roofline') or {}).get('schedule') or (b.get('schedule') or {})) if isinstance(b.get('roofline') or b.get('schedule'), dict) else {}
            if isinstance(sc, dict) and sc.get('hbm_frac_counter') is not None:
                sm[name].append(_r(sc['hbm_frac_counter']))
            for prec in ('f16x3', 'bf16'):
                alt = b.get('alt_precision_' + prec)
                if isinstance(alt, dict) and 'ms_per_step' in alt:
                    sm[name + '.' + prec] = [_r(alt['ms_per_step']), None, None]
            e3 = b.get('cfg3')            # inference_bf16: the configs[2] shape beside the north_star shape
            if isinstance(e3, dict) and 'ms_per_step' in e3:
                sm[name + '.cfg3'] = [_r(e3['ms_per_step']), _r(e3.get('hbm_frac')), _r(e3.get('mfma_frac'))]
            elif isinstance(e3, list) and len(e3) == 5:
                sm[name + '.cfg3'] = e3[2:]
    return sm


def write_detail(out):
    """The full-precision line with every per-kernel table next to the run (best effort; never fails the bench)."""
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, 'bench_detail.json')
        with open(path, 'w') as f:
            json.dump(out, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (cfg2: 32)')
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--algo', default='auto', choices=['auto', 'direct'])
    ap.add_argument('--no-alt', action='store_true', help="skip the additional measurements (f16x3 mode, cfg3, ResBlock1, RCCL sync overhead)")
    ap.add_argument('--precision', default='f32', choices=['f32', 'f16x3', 'bf16'],
                    help="precision mode of the TIMED steps (default: exact fp32; the others are for profiling that mode)")
    ap.add_argument('--resblock', default='2', choices=['1', '2'], help="'1': the ResBlock1 generator (h.resblock == '1'); default ResBlock2")
    ap.add_argument('--force-pg', action='store_true',
                    help='--gpus 1 only: create a ONE-rank RCCL process group and keep the CondBN statistics all-reduces inside the timed region')
    ap.add_argument('--verbose', action='store_true',
                    help='print the full line (per-kernel tables, launch tags, every sub-measurement); default: the compact line '
                         '(< 8 KB, ends in `summary`), the full one goes to gpurun_out/bench_detail.json')
    ap.add_argument('--dry-run', action='store_true',
                    help='first-contact check of the N-rank plumbing WITHOUT a GPU: launcher, rendezvous (set V2W_BENCH_BACKEND=gloo), per-rank '
                         'seeds and shards, barrier, max-over-ranks and the one JSON line are exercised; no forward runs and `value` is null')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the run asked for is not the run that was launched')
    if args.dry_run:
        raise SystemExit(dry_run(args, rank, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the Vec2Wav HIP path has no CPU fallback)')
    if os.environ.get('V2W_BENCH_DEVICE') is None and torch.cuda.device_count() < world:
        raise SystemExit(f'--gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible')
    # test hooks for a 1-GPU box: V2W_BENCH_DEVICE pins every rank to one device, V2W_BENCH_BACKEND=gloo replaces RCCL
    dev_index = int(os.environ.get('V2W_BENCH_DEVICE', local_rank))
    backend = os.environ.get('V2W_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    redirect = stdout_to_stderr()
    redirect.__enter__()          # everything until the JSON line (RCCL's banner included) goes to stderr
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif args.force_pg:
        single_rank_rccl_group(dev)
    grouped = world > 1 or args.force_pg

    from wavthruvec_pytorch_amd import Generator, synthetic, workmodel, hipops

    h = synthetic.make_hparams(num_wv_feat=768, resblock='1' if args.resblock == '1' else 1)
    B, T = args.batch, args.frames
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))
    g = g.to(dev).train()
    g.algo = hipops.ALGO_DIRECT if args.algo == 'direct' else hipops.ALGO_AUTO
    g.precision = args.precision
    if grouped:
        g.enable_sync_batchnorm(single_rank_collective=True if args.force_pg else None)
    x, spk, nz = synthetic.make_inputs(h, B, T, seed=1234 + rank, device=dev)
    inp = (x, spk, nz)
    up = synthetic.total_upsample(h)
    samples_per_step = world * B * T * up

    def barrier():
        if grouped:
            if backend == 'nccl':
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    def max_over_ranks(v):
        if world > 1:
            t = torch.tensor([v], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.item()
        return v

    # ---- the timed region
    elapsed, step_ms = run_steps(g, inp, args.steps, args.warmup, barrier)
    elapsed = max_over_ranks(elapsed)
    event_median_ms = max_over_ranks(median(step_ms)) if step_ms else None
    rccl_ranks = dist.get_world_size() if (grouped and backend == 'nccl') else (1 if world == 1 else None)
    extras = world == 1 and rank == 0 and args.algo == 'auto' and not args.no_alt and args.precision == 'f32' and args.resblock == '2' \
        and (B, T) == (32, 256)

    # ---- the same step with Generator.precision = 'f16x3' (wide Conv1d layers on the f16 matrix pipe with split operands,
    # fp32 accumulation; same parity bar).  Reported BESIDE the exact-fp32 `value`, never instead of it.
    alt = None
    if args.algo == 'auto' and not args.no_alt and args.precision == 'f32' and args.resblock == '2':
        with torch.no_grad():
            y32 = g(*inp).clone()
            g.precision = 'f16x3'
            alt_elapsed, _ms = run_steps(g, inp, args.steps, args.warmup, barrier)
            alt_elapsed = max_over_ranks(alt_elapsed)
            diff = max_over_ranks((g(*inp) - y32).abs().max().item())
            g.precision = 'f32'
            g(*inp)                   # back on the fp32 fragments for the profiling pass below
        alt = dict(precision='f16x3', value=samples_per_step * args.steps / alt_elapsed, unit='samples/s',
                   ms_per_step=alt_elapsed / args.steps * 1e3, max_abs_diff_vs_f32_path=diff, parity_bar=1e-4,
                   arithmetic='Conv1d layers with C_out >= 64 and the fused C = 32 / 16 residual stages: x = x_hi + x_lo (f16), '
                              'x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on v_mfma_f32_32x32x16_f16, fp32 accumulate; transposed convs, '
                              'BatchNorm and conv_post exact fp32')

    # ---- roofline of the dominant kernel: events around every conv launch, on the launching stream, live
    suffix = None
    if (B, T) == (32, 256) and args.resblock == '2':
        suffix = {'f32': '_cfg2_hbm_traffic.json', 'bf16': '_cfg2_bf16_hbm_traffic.json'}.get(args.precision)
    if (B, T) == (64, 512) and args.precision == 'bf16' and args.resblock == '2':
        suffix = '_cfg3_bf16_hbm_traffic.json'
    roof = roofline_block(g, h, inp, B, T, args.precision, args.algo, elapsed / args.steps, suffix, record=rank == 0)

    # ---- what the data-parallel statistics exchange costs per forward, on a one-rank RCCL group (the only RCCL run a one-GPU box allows)
    sync = None
    if extras or (args.force_pg and rank == 0 and world == 1):
        try:
            sync = stat_sync_overhead(g, inp, dev, max(5, args.steps // 2))
        except Exception as e:            # a box without a usable RCCL must not lose the bench line
            sync = dict(error=f'{type(e).__name__}: {e}')

    # ---- the ResBlock1 generator (h.resblock == '1', models.py:13-44) at the cfg2 shape: SURVEY 8(d) "ResBlock1 reported additionally"
    rb1 = None
    if extras:
        del g
        torch.cuda.empty_cache()
        h1 = synthetic.make_hparams(num_wv_feat=768, resblock='1')
        g1 = Generator(h1)
        g1.load_state_dict(synthetic.make_state_dict(h1, seed=0))
        g1 = g1.to(dev).train()
        n1 = max(5, args.steps // 2)
        el1, ms1 = run_steps(g1, inp, n1, max(2, args.warmup // 2))
        f1, b1 = workmodel.totals(h1, B, T, 4)
        s1 = el1 / n1
        r1 = roofline_block(g1, h1, inp, B, T, 'f32', 'auto', s1, None)
        rb1 = dict(workload=f'Generator.forward with ResBlock1 (h.resblock == \'1\'), B={B} x T={T}, 768-d, x320, train mode, exact fp32',
                   dtype='f32', steps=n1, ms_per_step=s1 * 1e3, ms_per_step_event_median=median(ms1), value=B * T * up / s1, unit='samples/s',
                   flops=f1, algorithmic_bytes=b1, mfma_frac=f1 / s1 / 1e12 / PEAK_FP32_MFMA_TFLOPS, hbm_frac=b1 / s1 / 1e9 / PEAK_HBM_GBS,
                   roofline={k: r1[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us', 'launches_per_step', 'per_kernel')},
                   parity='tests/test_hip_generator.py::test_generator_resblock1_full_size_vs_oracle_train (|dy| <= 1e-4 at this size)')
        del g1
        torch.cuda.empty_cache()

    # ---- the other single-GPU BASELINE configurations as first-class blocks (their own workload, byte count and dominant-kernel roofline):
    #   configs[2]  B = 64 x T = 512, bf16 compute / fp32 accumulate, bf16 activation storage          -> cfg3_bf16
    #   the north_star's literal shape B = 32 x T = 256 in that arithmetic                              -> cfg2_bf16
    #   configs[4]  1024-d latents, upsample (8,5,4,2,2) x640, B = 16 x T = 256, exact fp32             -> cfg5_f32
    # and the generator training step at the cfg2 shape                            -> train_step
    cfg3 = cfg2b = cfg5 = train = rb1b = infer = None
    if extras:
        def guarded(fn, *a, **kw):
            try:
                return fn(*a, **kw)
            except Exception as e:        # one extra block must not lose the bench line
                return dict(error=f'{type(e).__name__}: {e}')
        bf16_parity = ("no farther from the fp32 oracle than the reference's own bf16 autocast on the same inputs (max and rms): "
                       'tests/test_hip_generator.py::test_generator_cfg3_full_size_vs_oracle_train')
        cfg3 = guarded(config_block, dict(num_wv_feat=768), 64, 512, 'bf16', max(5, 5 * args.steps), max(8, args.warmup), dev,
                       'BASELINE configs[2]: Generator.forward, B=64 x T=512, 768-d latents, x320, train mode, bf16 compute / fp32 accumulate, '
                       'bf16 activation storage', bf16_parity, '_cfg3_bf16_hbm_traffic.json')
        cfg2b = guarded(config_block, dict(num_wv_feat=768), 32, 256, 'bf16', max(5, 5 * args.steps), max(8, args.warmup), dev,      # (1.4 ms steps: 100 of them,
                        # like cfg3 - over 20 the start / drain of the timed loop is 1-3 % of the reading)
                        'the north_star shape B=32 x T=256 (BASELINE configs[1]) in the configs[2] arithmetic: bf16 compute / fp32 accumulate, '
                        'bf16 activation storage, train mode', bf16_parity, '_cfg2_bf16_hbm_traffic.json')
        cfg5 = guarded(config_block, dict(num_wv_feat=1024, upsample_rates=[8, 5, 4, 2, 2], upsample_kernel_sizes=[16, 11, 8, 4, 4]), 16, 256, 'f32',
                       max(5, args.steps // 2), max(2, args.warmup // 2), dev,
                       'BASELINE configs[4]: 1024-d latents, upsample (8,5,4,2,2) x640, B=16 x T=256, train mode, exact fp32',
                       'tests/test_hip_generator.py::test_generator_cfg5_full_size_vs_oracle_train (|dy| <= 1e-4 at this size)')
        rb1b = guarded(config_block, dict(num_wv_feat=768, resblock='1'), 32, 256, 'bf16', max(5, args.steps // 2), max(2, args.warmup // 2), dev,
                       "Generator.forward with ResBlock1 (h.resblock == '1') at B=32 x T=256 in the configs[2] arithmetic: bf16 compute / fp32 "
                       'accumulate, bf16 tensors between all layers, train mode',
                       'tests/test_hip_generator.py::test_generator_resblock1_bf16_full_size_vs_oracle_train (the reference autocast bar)')
        infer = guarded(inference_block, dev, max(5, 5 * args.steps), max(8, args.warmup))
        train = guarded(train_step_block, dev, B, T, max(4, args.steps // 4))
        if isinstance(train, dict) and 'error' not in train:
            for prec in ('f16x3', 'bf16'):
                alt_t = guarded(train_step_block, dev, B, T, max(4, args.steps // 4), prec)
                train['alt_precision_' + prec] = {k: alt_t.get(k) for k in ('ms_per_step', 'value', 'tflops', 'workload', 'error') if k in alt_t}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(synthetic.make_hparams(num_wv_feat=768))

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        rbname = 'ResBlock1' if args.resblock == '1' else 'ResBlock2'
        out = {
            'metric': baseline_metric(),
            'value': samples_per_step * args.steps / elapsed,
            'unit': 'samples/s',
            'n_gpus': world, 'rccl_ranks': rccl_ranks, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'ms_per_step_event_median': event_median_ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'f16x3': 'f32 as f16 hi+lo (3 MFMA per product), f32 accumulate',
                      'bf16': 'bf16 operands, f32 accumulate'}[args.precision], 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[1]: B={B}/GPU x T={T} frames, 768-d latents, upsample (5,4,4,2,2) x320, '
                                   f'{rbname}, train-mode CondBN, {args.precision}', 'global_batch': B * world, 'frames': T,
                       'parallelism': (f'dp{world} (batch shards, RCCL all-reduce of CondBN stats)' if world > 1 else
                                       'single GPU' + (', one-rank RCCL group: the CondBN all-reduces are inside the timed region' if args.force_pg else ''))},
            'roofline': roof, 'cpu_baseline': cpu, 'alt_precision': alt, 'stat_sync': sync, 'resblock1_f32': rb1, 'cfg3_bf16': cfg3,
            'cfg2_bf16': cfg2b, 'inference_bf16': infer, 'cfg5_f32': cfg5, 'train_step': train, 'resblock1_bf16': rb1b,
        }
        blocks = ('alt_precision', 'cfg3_bf16', 'cfg2_bf16', 'inference_bf16', 'cfg5_f32', 'resblock1_f32', 'resblock1_bf16', 'train_step', 'stat_sync')
        if not args.verbose:
            detail = write_detail(out)
            out['roofline'] = compact_roofline(roof)
            out['cpu_baseline'] = compact_cpu(cpu)
            for name in blocks:
                out[name] = compact_block(out[name])
            out['detail'] = detail
        out['summary'] = summary_of(out, blocks)          # LAST key: [ms_per_step, hbm_frac, mfma_frac] of every block
    if dist.is_initialized():
        dist.destroy_process_group()
    redirect.__exit__()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
