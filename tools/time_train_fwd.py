#!/usr/bin/env python3
"""Times the forward half of a training step (activations saved) and the plain
forward on the default benchmark workload; GSAGE_LIB selects the library build."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphsage_amd as ga
from graphsage_amd import synthetic
from graphsage_amd.engine import SageForwardEngine
from graphsage_amd.train_engine import SageTrainEngine

B = int(os.environ.get('B', 4096))
dev = torch.device('cuda', 0)
g, u, it = synthetic.power_law_bipartite(1_000_000, 100_000, 50_000_000, 64, 64, alpha_item=1.1, seed=1234, device=dev)
hu, hi = synthetic.seed_edges(u, it, 1 << 20, seed=0)
gen = ga.GraphSAGEGenerator(g, B, [25, 10], head_node_types=['user', 'item'], seed=1)
model = ga.GraphSAGE([32, 32], gen, seed=11).build()
tr, fw = SageTrainEngine(g, model, B), SageForwardEngine(g, model, B)


def time_it(fn, n=50):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(5 + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def batch(i):
    lo = (i * B) % ((1 << 20) - B)
    return torch.cat([hu[lo:lo + B], hi[lo:lo + B]]).contiguous()


seeds = [batch(i) for i in range(64)]
wid = torch.arange(2 * B, dtype=torch.int64, device=dev)
print("lib %s  train-forward %.1f us  forward %.1f us" % (
    os.environ.get('GSAGE_LIB', 'default'),
    time_it(lambda i: tr.forward_train(seeds[i % 64], wid, rng_seed=7)),
    time_it(lambda i: fw.forward(seeds[i % 64], wid, rng_seed=7))))
