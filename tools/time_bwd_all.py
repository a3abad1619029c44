#!/usr/bin/env python3
"""Times the backward half of a training step on the default benchmark workload: the one-kernel form and the
separate launches; GSAGE_LIB selects the library build."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphsage_amd as ga
from graphsage_amd import synthetic
from graphsage_amd.train import Trainer, LinkHead

B = int(os.environ.get('B', 4096))
dev = torch.device('cuda', 0)
g, u, it = synthetic.power_law_bipartite(1_000_000, 100_000, 50_000_000, 64, 64, alpha_item=1.1, seed=1234, device=dev)
hu, hi = synthetic.seed_edges(u, it, 1 << 20, seed=0)
gen = ga.GraphSAGEGenerator(g, B, [25, 10], head_node_types=['user', 'item'], seed=1)
model = ga.GraphSAGE([32, 32], gen, seed=11).build()
tr = Trainer(model, head=LinkHead(32, dev, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=B)
seeds = torch.cat([hu[:B], hi[:B]]).contiguous()
wid = torch.arange(2 * B, dtype=torch.int64, device=dev)
y = (torch.arange(B, device=dev) % 2).float()
tr.step_device(seeds, y, wid, rng_seed=7)
eng, hs = tr.engine, tr._head_step


def bwd(_):
    eng.backward_head(tr.head.kernel, tr.head.bias, y, tr.head.kernel.grad, tr.head.bias.grad, hs.pred, hs.loss)


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


out = []
for fused in (True, False):
    eng.fused_backward = fused
    out.append(time_it(bwd))
print("lib %s  backward in one kernel %.1f us  separate launches %.1f us" % (os.environ.get('GSAGE_LIB', 'default'), out[0], out[1]))
