"""CPU restatement of the reference neighbor sampler -- test infrastructure.

``walk_sequential`` follows ``sampler.py:54-94`` step for step: one FIFO queue
per root, at depth d <= len(num_of_walks) the frontier node's neighbor list is
sampled ``num_of_walks[d-1]`` times with replacement through ``rs.choices``
(``:78``), an empty list raises ``ValueError`` (``:81-82``), and the returned
walk is ``[[root], hop1, hop2 of hop1[0], ..., hop3 of hop2[0][0], ...]``
(``:64,84``).

``TapeRS`` turns a pre-drawn uint32 tape into such an ``rs``: draw i of the
whole run takes word i and picks ``pop[(r * len(pop)) >> 32]``.  Because the
queue is FIFO, the words a walk consumes are laid out hop by hop in slot order,
which is the layout the device sampler indexes directly
(``tape[walk * T + hop_offset[h] + slot]``, ``T = sum of cumulative fan-outs``).

``walk_keyed`` is the same walk with the draw for (walk, hop, slot) supplied by
a function instead of a sequential stream: with ``tape_draw`` it must equal
``walk_sequential`` on the same tape, with ``philox.draw_u32`` it is the
statement of what the device does in its counter-RNG mode.
"""
from collections import deque

import numpy as np

from . import philox


class TapeRS:
    """Object with the one method the walker uses of ``random`` (``choices``),
    fed from a tape of pre-drawn 32-bit words."""

    def __init__(self, tape):
        self.tape = np.asarray(tape, dtype=np.uint32)
        self.pos = 0

    def choices(self, population, k=1):
        n = len(population)
        out = []
        for _ in range(k):
            r = int(self.tape[self.pos])
            self.pos += 1
            out.append(population[(r * n) >> 32])
        return out


def walk_sequential(graph, num_of_walks, nodes, rs):
    """``BreadthFirstWalker.run_breadth_first_walk`` (``sampler.py:47-96``)."""
    walk_length = len(num_of_walks)
    walks = []
    for node in nodes:
        walk = []
        queue = deque()
        node_type = graph.get_node_type(node)
        queue.extend([(node, node_type, 0)])
        walk.append([node])
        while len(queue) > 0:
            current_node, _current_type, depth = queue.popleft()
            depth = depth + 1
            if depth <= walk_length:
                neighbors = graph.get_neighbors_from_node(current_node)
                if len(neighbors) > 0:
                    samples = rs.choices(neighbors, k=num_of_walks[depth - 1])
                else:
                    raise ValueError("Samples must exist. Self-node also can be a sample")
                walk.append(samples)
                queue.extend([(s, graph.get_node_type(s), depth) for s in samples])
        walks.append(walk)
    return walks


def hop_sizes(num_of_walks):
    """Slots per hop: [n0, n0*n1, n0*n1*n2, ...]."""
    out, p = [], 1
    for n in num_of_walks:
        p *= int(n)
        out.append(p)
    return out


def tape_length(num_of_walks):
    return sum(hop_sizes(num_of_walks))


def tape_draw(tape, num_of_walks, first_walk=0):
    sizes = hop_sizes(num_of_walks)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    t = sum(sizes)
    tape = np.asarray(tape, dtype=np.uint32)

    def draw(walk, hop, slot):
        return int(tape[(walk - first_walk) * t + int(offs[hop]) + slot])
    return draw


def philox_draw(rng_seed):
    def draw(walk, hop, slot):
        return int(philox.draw_u32(walk, hop, slot, rng_seed))
    return draw


def walk_keyed(graph, num_of_walks, nodes, draw, first_walk=0):
    """Same walk as ``walk_sequential`` with draws addressed by (walk, hop, slot).

    Returns per-hop int64 arrays ``[len(nodes), hop_sizes[h]]`` (the dense form
    the device writes) instead of nested lists.
    """
    sizes = hop_sizes(num_of_walks)
    hops = [np.empty((len(nodes), s), dtype=np.int64) for s in sizes]
    for w, root in enumerate(nodes):
        frontier = [root]
        for h, k in enumerate(num_of_walks):
            nxt = []
            for j, node in enumerate(frontier):
                neighbors = graph.get_neighbors_from_node(node)
                if len(neighbors) == 0:
                    raise ValueError("Samples must exist. Self-node also can be a sample")
                n = len(neighbors)
                for i in range(k):
                    slot = j * k + i
                    r = draw(first_walk + w, h, slot)
                    nxt.append(neighbors[(r * n) >> 32])
            hops[h][w, :] = nxt
            frontier = nxt
    return hops


def walks_to_hops(walks, num_of_walks):
    """Nested reference walks -> per-hop dense arrays (BFS order is slot order)."""
    sizes = hop_sizes(num_of_walks)
    hops = [np.empty((len(walks), s), dtype=np.int64) for s in sizes]
    for w, walk in enumerate(walks):
        pos = 1
        groups = 1
        for h, k in enumerate(num_of_walks):
            flat = []
            for g in range(groups):
                flat.extend(walk[pos + g])
            pos += groups
            hops[h][w, :] = flat
            groups = sizes[h]
    return hops


def hops_to_walks(roots, hops, num_of_walks):
    """Per-hop dense arrays -> the nested list structure of ``sampler.py:64,84``."""
    walks = []
    for w, root in enumerate(roots):
        walk = [[int(root)]]
        for h, k in enumerate(num_of_walks):
            row = hops[h][w]
            for g in range(len(row) // k):
                walk.append([int(v) for v in row[g * k:(g + 1) * k]])
        walks.append(walk)
    return walks


def walk_vectorised(indptr, indices, roots, num_of_walks, rng_seed=None, first_walk=0, tape=None):
    """numpy-vectorised form of ``walk_keyed`` over a CSR, for inputs too large
    for Python loops.  Exactly one of ``rng_seed`` / ``tape`` is given.
    ``indices`` may also be any object whose ``[int64 array]`` returns the
    entries at those positions (attribute ``lazy`` set), for neighbor arrays
    kept elsewhere than host memory."""
    indptr = np.asarray(indptr)
    if not getattr(indices, 'lazy', False):
        indices = np.asarray(indices)
    roots = np.asarray(roots, dtype=np.int64)
    sizes = hop_sizes(num_of_walks)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    t = sum(sizes)
    nw = len(roots)
    walk_ids = first_walk + np.arange(nw, dtype=np.int64)
    frontier = roots[:, None]
    hops = []
    for h, k in enumerate(num_of_walks):
        parents = np.repeat(frontier, k, axis=1)                 # [nw, sizes[h]]
        start = indptr[parents]
        deg = indptr[parents + 1] - start
        if (deg == 0).any():
            raise ValueError("Samples must exist. Self-node also can be a sample")
        slot = np.arange(sizes[h], dtype=np.int64)[None, :]
        if tape is not None:
            r = np.asarray(tape, dtype=np.uint32).reshape(nw, t)[:, offs[h]:offs[h] + sizes[h]]
        else:
            r = philox.draw_u32(np.broadcast_to(walk_ids[:, None], parents.shape), h,
                                np.broadcast_to(slot, parents.shape), rng_seed)
        idx = philox.index_from_u32(r, deg)
        frontier = indices[start + idx].astype(np.int64)
        hops.append(frontier)
    return hops
