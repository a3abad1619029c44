"""Independent batches in flight: the one-kernel forward on one, two and three streams (each with its own
output rows).  (Timing aid.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
import graphsage_amd as ga                      # noqa: E402
from graphsage_amd import kernels, synthetic    # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
cfg, b = bench.WORKLOADS[name], 4096
dev = torch.device('cuda', 0)
g, u, it = synthetic.power_law_bipartite(cfg['nu'], cfg['ni'], cfg['e'], cfg['fu'], cfg['fi'], alpha_item=cfg['alpha_item'],
                                         alpha_user=cfg['alpha_user'], seed=1234, device=dev)
hu, hi = synthetic.seed_edges(u, it, 1 << 20, seed=0)
gen = ga.GraphSAGEGenerator(g, b, cfg['fan'], head_node_types=['user', 'item'], seed=7)
m = ga.GraphSAGE(cfg['layers'], gen, seed=7).build()
t = g.device_tensors()
weights = {(l, k): (m._aggs[l][k].w_self, m._aggs[l][k].w_neigh, m._aggs[l][k].bias) for l in (0, 1) for k in ('user', 'item')}
nb = 100
seeds = [torch.cat([hu[k * b:(k + 1) * b], hi[k * b:(k + 1) * b]]).contiguous() for k in range(nb)]
wid = torch.arange(2 * b, dtype=torch.int64, device=dev)
err = torch.zeros(2, dtype=torch.int32, device=dev)


def calls(n):
    return [kernels.Sage2Call(t['indptr'], t['indices'], t['user'], t['item'], cfg['fan'], weights, True, b) for _ in range(n)]


def run(label, cs, streams):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, s in enumerate(seeds):
            with torch.cuda.stream(streams[k % len(streams)]):
                cs[k % len(cs)].run(s, err, rng_seed=3, walk_ids=wid)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%-40s %7.1f us per batch" % (label, dt * 1e6 / nb), flush=True)
    return [c.emb.clone() for c in cs]


st = [torch.cuda.Stream() for _ in range(3)]
run("one stream", calls(1), st[:1])
run("two streams", calls(2), st[:2])
run("three streams", calls(3), st[:3])
print("error flags:", err.tolist())
