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
