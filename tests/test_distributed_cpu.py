"""Data-parallel CondBN on CPU (gloo, world_size 2): the product's shard + all-reduce helpers
(wavthruvec_pytorch_amd.distributed) wired into the oracle must reproduce the single-process run on the
concatenated global batch (SURVEY.md 8(e) parity definition).  The GPU path uses the same helpers with RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vec2wav_oracle as O
from wavthruvec_pytorch_amd import synthetic
from wavthruvec_pytorch_amd.distributed import BNStatSync, shard_batch, shard_bounds


def test_shard_bounds_cover_batch_exactly():
    for B in (1, 2, 7, 32, 33):
        for W in (1, 2, 3, 8):
            cuts = [shard_bounds(B, r, W) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            for (a, b), (c, d) in zip(cuts, cuts[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, T, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    h = synthetic.make_hparams(num_wv_feat=768)
    sd = synthetic.make_state_dict(h, seed=0)
    full = synthetic.make_inputs(h, B, T, seed=1234)
    mine = shard_batch(full, rank, world)
    sync = BNStatSync()

    def sync_stats(s, ss, n):
        # same wire format as the HIP path: [sum_c | sumsq_c | count] fp64, one all-reduce(sum)
        buf = torch.cat([s, ss, torch.tensor([float(n)], dtype=torch.float64)])
        sync(buf)
        C = s.numel()
        return buf[:C], buf[C:2 * C], buf[2 * C].item()

    y, nb = O.generator_forward(sd, h, *mine, training=True, sync_stats=sync_stats)
    torch.save({'y': y, 'nb': nb}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_forward_equals_global_batch(tmp_path):
    B, T, world = 3, 6, 2            # uneven shards: 2 + 1 samples
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, T, str(tmp_path)), nprocs=world, join=True)
    h = synthetic.make_hparams(num_wv_feat=768)
    sd = synthetic.make_state_dict(h, seed=0)
    full = synthetic.make_inputs(h, B, T, seed=1234)
    want, nb_want = O.generator_forward(sd, h, *full, training=True)
    parts = [torch.load(os.path.join(str(tmp_path), f'rank{r}.pt')) for r in range(world)]
    got = torch.cat([p['y'] for p in parts], dim=0)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-5
    for r in range(world):                       # every rank ends with the same (global) buffers
        for k, v in nb_want.items():
            g = parts[r]['nb'][k]
            if v.dtype == torch.long:
                assert g.item() == v.item(), k
            else:
                assert (g.float() - v.float()).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item()), k


def test_bn_stat_sync_requires_process_group():
    if dist.is_initialized():
        pytest.skip('a process group is already initialised in this interpreter')
    with pytest.raises(RuntimeError):
        BNStatSync()


# ---------------------------------------------------------------------------------------------------------------
# 8-GPU first contact, without the hardware
def _run(cmd, env_extra=None, timeout=600, cwd=None):
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=cwd)


@pytest.mark.timeout(600)
def test_bench_eight_rank_dry_run_over_gloo():
    """`python bench.py --gpus 8 --dry-run` on a box without GPUs: bench.py starts its 8 ranks itself (torch.distributed.run, 127.0.0.1),
    they meet over gloo, and rank 0 prints exactly ONE JSON line for the run that was asked for - n_gpus 8, global batch 256 (weak scaling,
    B = 32 per rank), eight different input seeds and eight disjoint shards that cover the global batch."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--dry-run'], {'V2W_BENCH_BACKEND': 'gloo', 'OMP_NUM_THREADS': '1'})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 256 and d['scaling'] == 'weak' and d['value'] is None
    ranks = d['dry_run']['ranks']
    assert [q['rank'] for q in ranks] == list(range(8))
    assert len({q['seed'] for q in ranks}) == 8 and len({q['x_checksum'] for q in ranks}) == 8
    cuts = [tuple(q['shard']) for q in ranks]
    assert cuts[0][0] == 0 and cuts[-1][1] == 256 and all(b == c for (_a, b), (c, _d) in zip(cuts, cuts[1:]))
    assert d['dry_run']['samples_per_step'] == 8 * 32 * 256 * 320
    # the run asked for is the run reported: a launcher that started another world size is refused
    bad = _run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--dry-run'],
               {'V2W_BENCH_BACKEND': 'gloo', 'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith('{')]


_BUILDER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
from wavthruvec_pytorch_amd import build as b
d = sys.argv[2]
b.CSRC = os.path.join(d, 'csrc'); b.OBJ_DIR = os.path.join(b.CSRC, '_obj'); b.LIB_PATH = os.path.join(d, 'libfake.so')
b.SOURCES = ['a.hip', 'b.hip', 'c.hip']; b.HEADERS = ['h.h']
print(b.build(jobs=2))
'''

_FAKE_HIPCC = r'''#!/usr/bin/env python3
import os, sys, time
out = sys.argv[sys.argv.index('-o') + 1]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'calls.log'), 'a') as f:
    f.write(('link ' if '-shared' in sys.argv else 'compile ') + os.path.basename(out).split('.tmp')[0] + '\n')
time.sleep(0.3)
open(out, 'w').write('x')
'''


@pytest.mark.timeout(300)
def test_eight_concurrent_builders_compile_once(tmp_path):
    """torchrun starts 8 ranks that all import the package after a source edit: exactly ONE of them builds (build.py's flock), the others
    wait and find the library fresh.  Run here on a stand-in source tree with a logging stand-in compiler: 3 compiles + 1 link in all."""
    import subprocess
    import sys
    import stat
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path)
    os.makedirs(os.path.join(d, 'csrc'))
    for f in ('a.hip', 'b.hip', 'c.hip', 'h.h'):
        open(os.path.join(d, 'csrc', f), 'w').write('// stand-in\n')
    hipcc = os.path.join(d, 'hipcc')
    open(hipcc, 'w').write(_FAKE_HIPCC)
    os.chmod(hipcc, os.stat(hipcc).st_mode | stat.S_IEXEC)
    script = os.path.join(d, 'builder.py')
    open(script, 'w').write(_BUILDER)
    env = dict(os.environ, HIPCC=hipcc)
    procs = [subprocess.Popen([sys.executable, script, root, d], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(8)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    assert all(o[0].strip().endswith('libfake.so') for o in outs)
    calls = open(os.path.join(d, 'calls.log')).read().split('\n')
    assert sorted(c for c in calls if c) == ['compile a.hip.o', 'compile b.hip.o', 'compile c.hip.o', 'link libfake.so'], calls
    # a source edit afterwards: one stale object is recompiled and the library relinked - once, whoever gets there first
    import time
    time.sleep(1.1)                                                    # (mtime granularity: the edit must be newer than the library)
    os.utime(os.path.join(d, 'csrc', 'b.hip'), None)
    time.sleep(0.1)
    procs = [subprocess.Popen([sys.executable, script, root, d], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(8)]
    assert all(p.wait(timeout=240) == 0 for p in procs)
    calls2 = [c for c in open(os.path.join(d, 'calls.log')).read().split('\n') if c][4:]
    assert sorted(calls2) == ['compile b.hip.o', 'link libfake.so'], calls2
