"""One rank of the data-parallel training rehearsal on a GPU (started by tests/test_zy_gpu_distributed.py, never
collected by pytest).  RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment; every rank uses
device 0 (the ranks of a one-GPU box share it, which is why the backend is gloo).

Each rank builds the same small graph, takes ITS shard of the links through ``flow(rank=, world_size=)`` and
runs ``--steps`` optimisation steps through the static plan.  A world of one takes batches ``world_equiv`` times
as large, i.e. the union of the shards.  Everything the test compares is written to ``--out``/r<rank>_<mode>.pt:
link positions per step, the hop-1 ids and the neighbor means the forward pass saved (functions of the walks and
the features only), loss, averaged gradients and the weights after every step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def toy(nu=400, ni=80, f=64):
    import graphsage_amd as ga
    rng = np.random.default_rng(0)
    u = np.concatenate([rng.integers(0, nu, 3000), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 3000), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, f)).astype(np.float32),
             'item': rng.standard_normal((ni, f)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))}, np.concatenate([e, e[::-1]], 1))
    links = [(int(a), int(b)) for a, b in zip(e[0, :1024], e[1, :1024])]
    targets = (np.arange(len(links)) % 2).astype(np.float32)
    return g, links, targets


def run_mode(mode, overlap, args, rank, world, dev):
    import graphsage_amd as ga
    from graphsage_amd.train import LinkHead, Trainer
    g, links, targets = toy()
    b = args.batch * (args.world_equiv // world)            # a world of one takes the union of the shards
    gen = ga.GraphSAGEGenerator(g, b, [6, 3], head_node_types=['user', 'item'], seed=3)
    # ranks start from different weights: the Trainer's broadcast must make them rank 0's
    model = ga.GraphSAGE([32, 32], gen, seed=100 + rank).build()
    tr = Trainer(model, head=LinkHead(32, dev, seed=200 + rank), engine='plan', graph=g, batch_size=b,
                 overlap=overlap, collective_at_world_one=(world == 1 and overlap != 'none'))
    eng = tr.engine
    assert eng is not None and eng.walk_train and eng.self_ids is not None
    seq = gen.flow(links, targets=targets, shuffle=True, seed=7, rank=rank, world_size=world)
    rec = {'batch': b, 'world': world, 'initial': tr.bucket.flat_params.detach().cpu().clone(), 'steps': []}
    for k in range(args.steps):
        seeds, walk_ids, key, y = seq.roots(k)
        y_dev = torch.from_numpy(np.asarray(y, dtype=np.float32)).to(dev)
        fn = tr.step_graph if mode == 'graph' else tr.step_device
        loss, pred = fn(seeds, y_dev, walk_ids, rng_seed=key)
        torch.cuda.synchronize()
        n0 = eng.fanouts[0]
        step = {'positions': torch.tensor(seq.indices[k * b:(k + 1) * b]),
                'loss': float(loss), 'pred': pred.detach().reshape(-1).cpu().clone(),
                # rows of the per-type tables: the B roots, then the B * n0 hop-1 heads of the OTHER side's walks
                'ids': {t: eng.self_ids[t].cpu().clone() for t in ('user', 'item')},
                'means': {t: eng.x_mean[0][t][:b * (1 + n0)].cpu().clone() for t in ('user', 'item')},
                'params': tr.bucket.flat_params.detach().cpu().clone()}
        if overlap == 'none':
            # the bucket holds the sum over ranks (the update kernel applies 1 / world as it reads)
            step['grads'] = (tr.bucket.flat / world).cpu().clone()
        rec['steps'].append(step)
    tr.flush()
    torch.cuda.synchronize()
    rec['final'] = tr.bucket.flat_params.detach().cpu().clone()
    assert eng.err_flag.tolist() == [0, 0]
    return rec


def time_pieces(args, rank, world, dev):
    """What the collective and the update add to a replayed step at the benchmark's batch size: microseconds of
    (stage + replayed graph), of the all-reduce of the flat bucket and of the update launch, each bracketed by
    device synchronisation.  The ranks share one GPU, so the replay itself is slower than on a GPU of its own."""
    import graphsage_amd as ga
    from graphsage_amd import synthetic
    from graphsage_amd.train import Trainer
    b = 4096
    graph, u, it = synthetic.power_law_bipartite(200_000, 20_000, 5_000_000, 64, 64, alpha_item=1.1, alpha_user=None,
                                                 seed=1234, device=dev)
    hu, hi = synthetic.seed_edges(u, it, 1 << 16, seed=0)
    gen = ga.GraphSAGEGenerator(graph, b, [25, 10], head_node_types=['user', 'item'], seed=5)
    model = ga.GraphSAGE([32, 32], gen, seed=7).build()
    tr = Trainer(model, lr=1e-2, engine='plan', graph=graph, batch_size=b)
    n_b = hu.numel() // b
    seeds = torch.cat([hu[:n_b * b].reshape(n_b, b), hi[:n_b * b].reshape(n_b, b)], 1).contiguous()
    eid = torch.arange(n_b * b, dtype=torch.int64, device=dev).reshape(n_b, b)
    wid = torch.cat([2 * eid, 2 * eid + 1], 1).contiguous()
    y = (torch.arange(b, device=dev) % 2).float()
    mine = list(range(rank, n_b, world))
    for i in range(5):
        tr.step_graph(seeds[mine[i % len(mine)]], y, wid[mine[i % len(mine)]], rng_seed=1)
    torch.cuda.synchronize()
    dist.barrier()
    n = 30
    t_replay = t_coll = t_adam = 0.0
    from graphsage_amd import kernels
    for i in range(n):
        k = mine[(5 + i) % len(mine)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernels.stage_batch(seeds[k], tr.engine.seeds_buf, wid[k], tr._walk_buf, y, tr._target_buf, 0.0, None)
        tr._graph.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        scale = tr.bucket.all_reduce_sum(tr.group)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tr.opt.step(grad_scale=scale)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        t_replay += t1 - t0
        t_coll += t2 - t1
        t_adam += t3 - t2
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(n):
        k = mine[(5 + i) % len(mine)]
        tr.step_graph(seeds[k], y, wid[k], rng_seed=1)
    torch.cuda.synchronize()
    whole = (time.perf_counter() - t0) / n
    return {'batch': b, 'world': world, 'backend': dist.get_backend(), 'bucket_bytes': tr.bucket.flat.numel() * 4,
            'replay_us': t_replay / n * 1e6, 'all_reduce_us': t_coll / n * 1e6, 'update_us': t_adam / n * 1e6,
            'step_us': whole * 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='links per step per rank at --world-equiv ranks')
    ap.add_argument('--world-equiv', type=int, default=2)
    ap.add_argument('--backend', default='gloo')
    ap.add_argument('--time', action='store_true')
    args = ap.parse_args()
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)          # a stall: say where, and end (the test's deadline is 150 s)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    if args.backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    try:
        modes = [('device', 'none'), ('graph', 'none'), ('graph', 'delayed')]
        if args.backend == 'nccl':
            modes = [('graph', 'capture'), ('graph', 'delayed')]
        for mode, overlap in modes:
            rec = run_mode(mode, overlap, args, rank, world, dev)
            torch.save(rec, os.path.join(args.out, 'w%d_r%d_%s_%s.pt' % (world, rank, mode, overlap)))
            print('[w%d r%d] %s/%s ok, losses %s' % (world, rank, mode, overlap,
                                                     ' '.join('%.5f' % s['loss'] for s in rec['steps'])), flush=True)
        if args.time:
            t = time_pieces(args, rank, world, dev)
            if rank == 0:
                with open(os.path.join(args.out, 'w%d_timing.json' % world), 'w') as fh:
                    json.dump(t, fh)
                print('[timing] ' + json.dumps(t), flush=True)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
