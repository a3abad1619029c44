#!/usr/bin/env python3
"""Replays the captured training step back to back, without staging new inputs in between (for a kernel trace:
what lies between two replays is then the graph launch alone)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphsage_amd as ga
from graphsage_amd import synthetic
from graphsage_amd.train import Trainer, LinkHead

B = 4096
dev = torch.device('cuda', 0)
g, u, it = synthetic.power_law_bipartite(1_000_000, 100_000, 50_000_000, 64, 64, alpha_item=1.1, seed=1234, device=dev)
hu, hi = synthetic.seed_edges(u, it, 1 << 20, seed=0)
gen = ga.GraphSAGEGenerator(g, B, [25, 10], head_node_types=['user', 'item'], seed=1)
model = ga.GraphSAGE([32, 32], gen, seed=11).build()
tr = Trainer(model, head=LinkHead(32, dev, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=B)
seeds = torch.cat([hu[:B], hi[:B]]).contiguous()
wid = torch.arange(2 * B, dtype=torch.int64, device=dev)
y = (torch.arange(B, device=dev) % 2).float()
for _ in range(3):
    tr.step_graph(seeds, y, wid, rng_seed=7)
torch.cuda.synchronize()
for _ in range(20):
    tr._graph.replay()
torch.cuda.synchronize()
