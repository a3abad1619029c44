#!/usr/bin/env python3
"""Times the training step (sample + forward + backward + gradient all-reduce +
Adam) on the bench workload; prints seed edges/s."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import graphsage_amd as ga
from graphsage_amd import synthetic
from graphsage_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--warmup', type=int, default=5)
ap.add_argument('--batch', type=int, default=4096)
ap.add_argument('--workload', default='cfg2')
ap.add_argument('--engine', default='auto')
args = ap.parse_args()
cfg = B.WORKLOADS[args.workload]
dev = torch.device('cuda', 0)
graph, u, it = synthetic.power_law_bipartite(cfg['nu'], cfg['ni'], cfg['e'], cfg['fu'], cfg['fi'],
                                             alpha_item=cfg['alpha_item'], alpha_user=cfg['alpha_user'], seed=1234, device=dev)
hu, hi = synthetic.seed_edges(u, it, 1 << 18, seed=0)
links = torch.stack([hu, hi], 1).cpu().numpy()
gen = ga.GraphSAGEGenerator(graph, args.batch, cfg['fan'], head_node_types=['user', 'item'], seed=1)
model = ga.GraphSAGE(cfg['layers'], gen, seed=7)
tr = Trainer(model, engine=args.engine) if 'engine' in Trainer.__init__.__code__.co_varnames else Trainer(model)
import numpy as np
targets = (np.arange(len(links)) % 2).astype(np.float32)
seq = gen.flow(links, targets=targets, shuffle=False)
def step(i):
    batch, t = seq[i % len(seq)]
    return tr.step(batch, t)
for i in range(args.warmup): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(args.steps): step(args.warmup + i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"train_seed_edges_per_s": args.steps * args.batch / dt, "ms_per_step": dt / args.steps * 1e3}))
