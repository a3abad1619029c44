"""Breadth-first uniform neighbor sampler, executed by the HIP sampler kernel.

Interface of the reference's ``BreadthFirstWalker`` (``sampler.py:26-96``):
``BreadthFirstWalker(Graph, num_of_walks)`` (``TypeError`` unless a list,
``:36-37``) and ``run_breadth_first_walk(nodes)`` returning, per root, the list
``[[root], hop1, hop2 of hop1[0], ...]`` (``:64,84``).

Randomness.  The reference draws from the process-wide stdlib generator
(``sampler.py:24,40,78``).  Here every draw is addressed by
``(walk id, hop, slot)``:

* default: Philox4x32-10 keyed by ``seed``; walk ids continue from one call to
  the next (``next_walk_id``), so repeated calls give fresh samples and a run is
  reproducible from its seed;
* replay: ``set_replay(tape)`` supplies the 32-bit word of every draw of the
  next call in the order the reference would consume them, and the sampled
  neighbor is ``neighbors[(word * degree) >> 32]``.  Feeding the reference the
  same words yields the same walks bit for bit.
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
        self.next_walk_id = 0
        self._replay = None

    def __repr__(self):
        return "<Breadth First Walker with {} number of walks>".format(self.num_of_walks)

    def set_replay(self, tape):
        """Pre-drawn uint32 words for the next call (None: back to Philox)."""
        self._replay = None if tape is None else np.ascontiguousarray(tape, dtype=np.uint32)

    # -- device-side entry: everything stays on the GPU ---------------------
    def sample_device(self, seeds, walk_ids=None, check=True, rng_seed=None):
        """seeds: int32 GPU tensor of roots.  Returns the per-hop int32 tensors
        ``[len(seeds), n0*...*nh]`` in breadth-first order.  ``rng_seed`` keys
        this call's draws instead of the walker's own seed (which is left as it
        is)."""
        import torch
        dev = self.graph.device_tensors()
        replay = None
        if self._replay is not None:
            replay = torch.from_numpy(self._replay.view(np.int32)).to(seeds.device)
            self._replay = None
        hops, err = kernels.sample(dev['indptr'], dev['indices'], seeds, self.num_of_walks,
                                   rng_seed=self.seed if rng_seed is None else int(rng_seed),
                                   walk_id_base=self.next_walk_id,
                                   walk_ids=walk_ids, replay=replay)
        if walk_ids is None:
            self.next_walk_id += int(seeds.numel())
        if check:
            self.raise_on_error(err)
        return hops

    @staticmethod
    def raise_on_error(err_flag):
        """Turns the kernel's error words into the reference's exceptions (one
        device-to-host copy of 8 bytes; synchronises the stream)."""
        isolated, bad_seed = (int(v) for v in err_flag.tolist())
        if bad_seed:
            raise KeyError("a root node id supplied to the sampler is not in the graph")
        if isolated:
            raise IsolatedNodeError(
                "Samples must exist. Self-node also can be a sample (node %d has no neighbors)" % (isolated - 1))

    # -- reference-shaped entry ---------------------------------------------------
    def run_breadth_first_walk(self, nodes=None):
        import torch
        from ._lib import require_gpu
        nodes = list(nodes)
        for n in nodes:
            self.graph.get_node_type(n)            # KeyError for unknown nodes, as sampler.py:60
        if len(nodes) == 0:
            return []
        dev = require_gpu()
        seeds = torch.tensor(nodes, dtype=torch.int32, device=dev)
        hops = [h.cpu().numpy() for h in self.sample_device(seeds)]
        walks = []
        for w, root in enumerate(nodes):
            walk = [[root]]
            for h, k in enumerate(self.num_of_walks):
                row = hops[h][w].tolist()
                walk.extend(row[g:g + k] for g in range(0, len(row), k))
            walks.append(walk)
        return walks
