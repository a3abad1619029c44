#!/usr/bin/env python3
"""Host time to enqueue a training step of the benchmark shape against the step's time on the GPU: the step is
launched directly (Trainer.step_auto), which only works while the host stays ahead."""
import os, sys, time, torch
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
seeds = [torch.cat([hu[i * B:(i + 1) * B], hi[i * B:(i + 1) * B]]).contiguous() for i in range(16)]
wid = torch.arange(2 * B, dtype=torch.int64, device=dev)
y = (torch.arange(B, device=dev) % 2).float()
for i in range(10):
    tr.step_auto(seeds[i % 16], y, wid, rng_seed=7)
torch.cuda.synchronize()
import gc; gc.disable()
n = 300
t0 = time.perf_counter()
for i in range(n):
    tr.step_auto(seeds[i % 16], y, wid, rng_seed=7)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
