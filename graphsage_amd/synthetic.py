"""Synthetic user-item graphs built on the GPU (benchmark inputs).

The reference builds its adjacency with one ``np.where`` scan per node
(``bipartite_graph.py:53-60``) from an edge list that already holds both
directions (``train_movielens.py:130-132``); at tens of millions of edges that
is replaced here by: draw endpoints on the device and hand both columns to the
library's adjacency build (``gs_csr_build``: both directions, stable by source).
The result is the same CSR the ``Graph`` constructor would make from those edges.

Endpoint laws: users uniform; items Zipf(alpha) over item rank (rank 0 most
popular), truncated to the number of items.  One extra edge per user
(user u -- item u mod NI) guarantees every user and every item a neighbor,
because the sampler, like the reference, refuses nodes without one.
"""
import numpy as np
import torch

from .bipartite_graph import Graph


def zipf_cdf(n, alpha, device):
    w = torch.arange(1, n + 1, dtype=torch.float64, device=device).pow_(-alpha)
    cdf = torch.cumsum(w, 0)
    return (cdf / cdf[-1]).to(torch.float32)


def power_law_bipartite(num_users, num_items, num_edges, feat_user, feat_item, alpha_item=1.1,
                        alpha_user=None, seed=1234, device=None, chunk=1 << 26):
    """Returns (Graph with device-resident CSR/features, user endpoints, item
    endpoints of the undirected edges as int32 GPU tensors, item ids global)."""
    device = device or torch.device('cuda', torch.cuda.current_device())
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    n_rand = num_edges - num_users
    if n_rand < 0:
        raise ValueError("need at least one edge per user")

    def draw(n_nodes, alpha, count):
        if alpha is None:
            return torch.randint(0, n_nodes, (count,), generator=gen, device=device, dtype=torch.int32)
        cdf = zipf_cdf(n_nodes, alpha, device)
        out = torch.empty(count, dtype=torch.int32, device=device)
        for a in range(0, count, chunk):
            u = torch.rand(min(chunk, count - a), generator=gen, device=device)
            out[a:a + u.numel()] = torch.searchsorted(cdf, u).clamp_(max=n_nodes - 1).to(torch.int32)
        return out

    forced_u = torch.arange(num_users, dtype=torch.int32, device=device)
    u = torch.cat([forced_u, draw(num_users, alpha_user, n_rand)])
    it = torch.cat([forced_u % num_items, draw(num_items, alpha_item, n_rand)]) + num_users
    if num_items > num_users:
        raise ValueError("the forced edges cover every item only when num_users >= num_items")

    fgen = torch.Generator(device=device)
    fgen.manual_seed(seed + 1)
    feats = {'user': torch.randn((num_users, feat_user), generator=fgen, device=device, dtype=torch.float32),
             'item': torch.randn((num_items, feat_item), generator=fgen, device=device, dtype=torch.float32)}
    g = Graph.from_edges_device(feats, u, it, symmetric=True)
    return g, u, it


def seed_edges(u, it, count, seed=0):
    """The first ``count`` undirected edges in an order shuffled by
    ``numpy.random.default_rng(seed)``: (user heads, item heads) int32 GPU tensors."""
    count = min(count, u.numel())
    perm = torch.from_numpy(np.random.default_rng(seed).permutation(count)).to(u.device)
    return u[:count][perm].contiguous(), it[:count][perm].contiguous()
