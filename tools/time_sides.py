"""How long do the user-rooted and the item-rooted walks of a cfg-2 batch take on their own?  Times the
one-kernel forward on a full batch, then with one side's roots replaced by -1 (walks without rows).
(Timing aid; the error flag is set by the -1 roots on purpose.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
import graphsage_amd as ga                      # noqa: E402
from graphsage_amd import synthetic             # noqa: E402
from graphsage_amd.engine import SageForwardEngine   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
cfg, b = bench.WORKLOADS[name], 4096
dev = torch.device('cuda', 0)
g, u, it = synthetic.power_law_bipartite(cfg['nu'], cfg['ni'], cfg['e'], cfg['fu'], cfg['fi'], alpha_item=cfg['alpha_item'],
                                         alpha_user=cfg['alpha_user'], seed=1234, device=dev)
hu, hi = synthetic.seed_edges(u, it, 1 << 20, seed=0)
gen = ga.GraphSAGEGenerator(g, b, cfg['fan'], head_node_types=['user', 'item'], seed=7)
model = ga.GraphSAGE(cfg['layers'], gen, seed=7).build()
eng = SageForwardEngine(g, model, b)
nb = 40
seeds = [torch.cat([hu[k * b:(k + 1) * b], hi[k * b:(k + 1) * b]]).contiguous() for k in range(nb)]
wid = torch.arange(2 * b, dtype=torch.int64, device=dev)


def run(which):
    batches = []
    for s in seeds:
        s = s.clone()
        if which == 'users':
            s[b:] = -1
        elif which == 'items':
            s[:b] = -1
        batches.append(s)
    for s in batches[:10]:
        eng.forward(s, wid, rng_seed=3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in batches:
        eng.forward(s, wid, rng_seed=3)
    e1.record()
    torch.cuda.synchronize()
    print("%-6s %7.1f us per step" % (which, e0.elapsed_time(e1) * 1e3 / nb), flush=True)


for which in ('both', 'users', 'items', 'both'):
    run(which)
