#!/usr/bin/env python3
"""Two-rank rehearsal of the data-parallel training step (run under
torch.distributed.run).  --backend gloo lets both ranks share one GPU."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument('--backend', default='gloo')
ap.add_argument('--mode', default='graph')
args = ap.parse_args()
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
def say(*a):
    print('[r%d]' % rank, *a, flush=True)
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
torch.cuda.set_device(dev)
dist.init_process_group(args.backend)
say('init ok')
import graphsage_amd as ga
from graphsage_amd.train import Trainer, LinkHead
rng = np.random.default_rng(0)
nu, ni, f, b, ns = 400, 80, 64, 64, [6, 3]
u = np.concatenate([rng.integers(0, nu, 3000), np.arange(nu)])
it = np.concatenate([rng.integers(0, ni, 3000), np.arange(nu) % ni]) + nu
e = np.stack([u, it])
feats = {'user': rng.standard_normal((nu, f)).astype(np.float32), 'item': rng.standard_normal((ni, f)).astype(np.float32)}
g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))}, np.concatenate([e, e[::-1]], 1))
gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
model = ga.GraphSAGE([32, 32], gen, seed=100 + rank).build()          # ranks start from different weights
tr = Trainer(model, head=LinkHead(32, dev, seed=200 + rank), engine='plan', graph=g, batch_size=b)
say('trainer ok')
flat0 = tr.bucket.flat_params.clone()
gathered = [torch.zeros_like(flat0) for _ in range(world)]
dist.all_gather(gathered, flat0)
assert all(torch.equal(gathered[0], x) for x in gathered), 'broadcast of initial weights failed'
links = np.stack([e[0, :b * world * 4], e[1, :b * world * 4]], 1)
targets = (np.arange(len(links)) % 2).astype(np.float32)
for step in range(4):
    idx = np.arange(step * b * world + rank, (step + 1) * b * world, world)          # rank-strided shard
    part = links[idx]
    seeds = torch.from_numpy(np.concatenate([part[:, 0], part[:, 1]]).astype(np.int32)).to(dev)
    wid = torch.from_numpy(np.concatenate([2 * idx, 2 * idx + 1]).astype(np.int64)).to(dev)
    t = torch.from_numpy(targets[idx]).to(dev)
    fn = tr.step_graph if args.mode == 'graph' else tr.step_device
    loss, _ = fn(seeds, t, wid, rng_seed=5)
    torch.cuda.synchronize()
    say('step', step, 'loss', float(loss))
flat = tr.bucket.flat_params.clone()
gathered = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
assert all(torch.equal(gathered[0], x) for x in gathered), 'ranks diverged'
assert not torch.equal(flat, flat0)
say('OK: ranks hold identical weights after 4 steps')
dist.destroy_process_group()
