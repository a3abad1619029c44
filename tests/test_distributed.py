"""Data-parallel plumbing under a 2-process gloo group on the CPU (no GPU): the
flat gradient bucket is averaged across ranks by one all-reduce, parameters are
broadcast from rank 0, link shards partition every epoch, and Adam follows the
Keras update rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graphsage_amd.generator import LinkSequence
from graphsage_amd.train import Adam, FlatGradBucket


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # ranks start from different weights
        params = [torch.randn(4, 3, requires_grad=True), torch.randn(5, requires_grad=True)]
        # what Trainer.broadcast_parameters does
        with torch.no_grad():
            flat = torch.cat([p.reshape(-1) for p in params])
            dist.broadcast(flat, src=0)
            params[0].copy_(flat[:12].view(4, 3))
            params[1].copy_(flat[12:])
        bucket = FlatGradBucket(params)
        opt = Adam(params, bucket.flat, lr=1e-2)
        ids = [(i, 100 + i) for i in range(23)]
        seq = LinkSequence(lambda heads, bn: heads, 4, ids, targets=list(range(23)), shuffle=True, seed=7,
                           rank=rank, world_size=world)
        seen = []
        for step in range(3):
            bucket.zero()
            heads, t = seq[step % len(seq)]
            seen.extend(int(v) for v in t)
            # a loss whose gradient depends on the rank's shard
            w = float(sum(t)) + 1.0
            loss = w * (params[0] ** 2).sum() + (params[1] * (rank + 1)).sum()
            loss.backward()
            assert params[0].grad.data_ptr() == bucket.flat.data_ptr()      # grads live in the bucket
            local = bucket.flat.clone()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            bucket.all_reduce_mean()
            assert torch.allclose(bucket.flat, sum(gathered) / world, atol=1e-6)
            opt.step()
        final = torch.cat([p.detach().reshape(-1) for p in params])
        both = [torch.zeros_like(final) for _ in range(world)]
        dist.all_gather(both, final)
        assert torch.equal(both[0], both[1]), "ranks diverged"
        torch.save({'final': final, 'indices': list(seq.indices)}, os.path.join(out_dir, 'r%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_average_and_shards(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'r0.pt', weights_only=True)
    r1 = torch.load(tmp_path / 'r1.pt', weights_only=True)
    assert torch.equal(r0['final'], r1['final'])
    assert sorted(r0['indices'] + r1['indices']) == list(range(23))


def test_adam_matches_keras_update_rule():
    p = torch.tensor([1.0, -2.0, 3.0], requires_grad=True)
    bucket = FlatGradBucket([p])
    opt = Adam([p], bucket.flat, lr=1e-2)
    m = np.zeros(3)
    v = np.zeros(3)
    x = np.array([1.0, -2.0, 3.0])
    for t in range(1, 6):
        bucket.zero()
        (p ** 3).sum().backward()
        g = 3 * x ** 2
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        x = x - 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)
        opt.step()
        np.testing.assert_allclose(p.detach().numpy(), x, rtol=1e-5)


def _run_bench(argv, env_extra=None, drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK')):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + argv, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_starts_its_own_ranks():
    """``bench.py --gpus 2`` with no launcher around it starts two ranks that meet
    (gloo here, no GPU work) and rank 0's line says how many it saw."""
    import json
    res = _run_bench(['--gpus', '2', '--backend', 'gloo', '--dry-launch'])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and lines[0]['ranks_seen'] == 2 and lines[0]['n_gpus'] == 2


def test_bench_refuses_a_rank_count_other_than_asked():
    res = _run_bench(['--gpus', '2', '--backend', 'gloo', '--dry-launch'],
                     env_extra={'WORLD_SIZE': '1', 'RANK': '0'}, drop=())
    assert res.returncode != 0 and 'ranks' in res.stderr
