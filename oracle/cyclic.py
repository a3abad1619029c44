"""CPU restatement of the reference's alternate walker -- test infrastructure.

``walk_cyclic`` follows ``backup/sampler_backup.py:47-114`` statement by statement:
per frontier node the next ``k`` entries of its neighbor list from a per-node
cursor (``graph.sampling_index_dict``, ``bipartite_graph.py:28,44-46``); while the
list has fewer than ``k`` entries left, what is left is taken, the LOCAL list is
replaced by a permutation of itself (``:87``; the stored list stays as it is), and
the cursor restarts at 0 (``:85-96``).  ``permute(list) -> list`` stands for
``numpy.random.permutation(...).tolist()``.

``TapePerm`` replays index permutations (``permute(x) = [x[j] for j in p]``);
``FeistelPerm`` is the counter-based permutation the device draws when no tape is
given: permutation number t of a run is a four-round Feistel network over the next
even power of two >= len(x), keyed by Philox4x32-10(counter = (t, round,
0x43594331), key = seed) and cycle-walked into range.  Pinning: the walk is pinned
by fixtures produced with the reference's own ``backup/sampler_backup.py``
(``tests/golden/make_golden.py``, ``cyclic_cases``); the Feistel generator has no
counterpart in the reference.
"""
from collections import deque

import numpy as np

from . import philox


class TapePerm:
    def __init__(self, perms):
        self.perms = [np.asarray(p, dtype=np.int64) for p in perms]
        self.pos = 0

    def __call__(self, x):
        p = self.perms[self.pos]
        self.pos += 1
        assert len(p) == len(x), "replayed permutation has the wrong length"
        return [x[int(j)] for j in p]


def feistel_index(t, length, i, seed):
    """Image of position ``i`` under permutation number ``t`` of a list of ``length`` entries."""
    bits = 2
    while (1 << bits) < length:
        bits += 2
    half = bits // 2
    mask = (1 << half) - 1
    rk = [int(philox.draw_u32(t, r, 0x43594331, seed)) for r in range(4)]
    x = i
    while True:
        l, r = x >> half, x & mask
        for rd in range(4):
            f = ((r ^ rk[rd]) * 0x9E3779B1) & 0xFFFFFFFF
            f ^= f >> 15
            f = (f * 0x85EBCA77) & 0xFFFFFFFF
            f ^= f >> 13
            l, r = r, l ^ (f & mask)
        x = (l << half) | r
        if x < length:
            return x


class FeistelPerm:
    def __init__(self, seed, first=0):
        self.seed = seed
        self.t = first

    def __call__(self, x):
        n = len(x)
        out = [x[feistel_index(self.t, n, i, self.seed)] for i in range(n)]
        self.t += 1
        return out


def walk_cyclic(graph, num_of_walks, nodes, cursor, permute):
    """``cursor``: dict node -> position, updated in place (the reference's
    ``graph.sampling_index_dict``)."""
    walk_length = len(num_of_walks)
    walks = []
    for node in nodes:
        walk = []
        queue = deque()
        queue.extend([(node, graph.get_node_type(node), 0)])
        walk.append([node])
        while len(queue) > 0:
            current_node, _type, depth = queue.popleft()
            depth = depth + 1
            if depth <= walk_length:
                neighbors = graph.get_neighbors_from_node(current_node)
                if len(neighbors) > 0:
                    index = cursor[current_node]
                    n = len(neighbors)
                    leftover = num_of_walks[depth - 1]
                    samples = []
                    while n - index < leftover:
                        samples.extend(neighbors[index:n])
                        neighbors = permute(neighbors)
                        leftover -= n - index
                        cursor[current_node] = 0
                        index = cursor[current_node]
                    samples.extend(neighbors[index:index + leftover])
                    walk.append(samples)
                    cursor[current_node] += leftover
                else:
                    raise ValueError("Samples must exist. Self-node also can be a sample")
                queue.extend([(s, graph.get_node_type(s), depth) for s in samples])
        walks.append(walk)
    return walks
