"""The reference's alternate walker, ``backup/sampler_backup.py``: neighbors taken
cyclically from a per-node cursor, without replacement until the list wraps.

Same interface as that file's ``BreadthFirstWalker`` (``sampler_backup.py:26-45``:
``TypeError`` unless ``num_of_walks`` is a list) and the same walk lists from
``run_breadth_first_walk`` (``:47-114``).  The per-node cursors are the graph's
``sampling_index_dict`` (``bipartite_graph.py:28,44-46``); here they live on the
GPU (``Graph.sampling_cursor()``) and carry over from call to call as the dict
does.  When a cursor runs off the end of a neighbor list the reference continues
in ``numpy.random.permutation(neighbors)`` (``:85-90``); here that permutation is
either replayed (``set_replay``: the index permutations numpy returned, in call
order -- feeding the reference the same ones gives the same walks bit for bit) or
drawn from the counter-based generator (Philox-keyed Feistel network, reproducible
from ``seed``).

The visits are sequential by the reference's own definition (a cursor depends on
every earlier visit of its node), so ``gs_sample_cyclic`` walks them with one
wavefront.  It is the variant for reproducing that sampler; nothing imports it in
the reference, and ``graphsage_amd.sampler.BreadthFirstWalker`` is the fast one.
"""
import numpy as np

from . import kernels
from .bipartite_graph import IsolatedNodeError

__all__ = ['BreadthFirstWalker']


class BreadthFirstWalker:
    def __init__(self, Graph, num_of_walks=None, seed=None):
        if not isinstance(num_of_walks, list):
            raise TypeError("num_of_walks must be list of integers")
        self.graph = Graph
        self.num_of_walks = num_of_walks
        self.walk_length = len(num_of_walks)
        self.seed = 0x9E3779B97F4A7C15 if seed is None else int(seed)
        self._state = None               # device int64[2]: permutations drawn, tape words used
        self._tape = None

    def __repr__(self):
        return "<Breadth First Walker with {} number of walks>".format(self.num_of_walks)

    def set_replay(self, permutations):
        """Index permutations for the coming calls, in the order the reference would
        ask numpy for them (None: back to the counter-based generator).  Each entry
        p stands for ``permutation(x) == x[p]``."""
        if permutations is None:
            self._tape = None
        else:
            flat = [np.asarray(p, dtype=np.int32).reshape(-1) for p in permutations]
            self._tape = np.concatenate(flat) if flat else np.zeros(0, dtype=np.int32)
        if self._state is not None:
            self._state[1] = 0                              # a new tape is read from its start

    def sample_device(self, seeds, check=True):
        """seeds: int32 GPU tensor of roots -> per-hop int32 tensors, breadth-first order."""
        import torch
        dev = self.graph.device_tensors()
        if self._state is None:
            self._state = torch.zeros(2, dtype=torch.int64, device=seeds.device)
        tape, tape_len = None, None
        if self._tape is not None:                          # (an empty record still means "replay": one spare word gives it an address)
            tape_len = int(self._tape.shape[0])
            tape = torch.from_numpy(np.concatenate([self._tape, np.zeros(1, np.int32)])).to(seeds.device)
        hops, err = kernels.sample_cyclic(dev['indptr'], dev['indices'], seeds, self.num_of_walks,
                                          self.graph.sampling_cursor(), self._state, rng_seed=self.seed, perm_tape=tape,
                                          perm_tape_len=tape_len)
        if check:
            isolated, other = (int(v) for v in err.tolist())
            if other == 1:
                raise KeyError("a root node id supplied to the sampler is not in the graph")
            if other == 2:
                raise ValueError("the replayed permutations ran out")
            if isolated:
                raise IsolatedNodeError(
                    "Samples must exist. Self-node also can be a sample (node %d has no neighbors)" % (isolated - 1))
        return hops

    def run_breadth_first_walk(self, nodes=None):
        import torch
        from ._lib import require_gpu
        nodes = list(nodes)
        for n in nodes:
            self.graph.get_node_type(n)                     # KeyError for unknown nodes, as sampler_backup.py:60
        if len(nodes) == 0:
            return []
        dev = require_gpu()
        hops = [h.cpu().numpy() for h in self.sample_device(torch.tensor(nodes, dtype=torch.int32, device=dev))]
        walks = []
        for w, root in enumerate(nodes):
            walk = [[root]]
            for h, k in enumerate(self.num_of_walks):
                row = hops[h][w].tolist()
                walk.extend(row[g:g + k] for g in range(0, len(row), k))
            walks.append(walk)
        return walks
