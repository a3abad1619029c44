"""CPU restatement of the reference batch generator -- test infrastructure.

* ``sampling_schema`` / ``type_adjacency_list``: ``generator.py:239-244`` and
  ``:339-345`` are written out by hand for exactly two hops.  Here they are
  produced for any depth by the rule those literals follow: slot ``2*h + side``
  holds hop ``h`` of the walks rooted at head ``side`` (0 = user head, 1 = item
  head), its node type alternates with ``h`` because the graph is bipartite, it
  collects walk-list entries ``[first(h), first(h) + groups(h))`` of every walk,
  and its only neighbor slot is ``2*(h+1) + side``.  For two hops this gives the
  reference's literals verbatim (checked in the tests); three hops is an
  extrapolation that no reference fixture covers.
* ``sample_features_faithful``: ``generator.py:281-337`` statement by
  statement, including the ``reduce(operator.concat, ..., [])`` regroup whose
  cost is quadratic in batch * fan-out (``:318-322,331``).  This is the form the
  CPU baseline times.
* ``sample_features_fast``: same result through per-hop dense arrays.
* ``OracleLinkSequence``: ``generator.py:17-110`` (batch slicing, shuffle through
  ``random.Random(seed).shuffle`` or the global ``random``).
"""
import operator
import random as _stdlib_random
from functools import reduce

import numpy as np

from . import walker as _walker

TYPES = ('user', 'item')


def sampling_schema(num_samples):
    n_hops = len(num_samples)
    sizes = [1] + _walker.hop_sizes(num_samples)          # nodes per hop incl. root
    first = [0, 1]
    for h in range(2, n_hops + 1):
        first.append(first[-1] + sizes[h - 2])             # groups of hop h-1 = sizes[h-2]
    groups = [1, 1] + [sizes[h - 1] for h in range(2, n_hops + 1)]
    schema = []
    for side in range(2):
        row = []
        for h in range(n_hops + 1):
            for s in range(2):
                node_type = TYPES[(s + h) % 2]
                idx = list(range(first[h], first[h] + groups[h])) if s == side else []
                row.append((node_type, idx))
        schema.append(row)
    return schema


def type_adjacency_list(n_hops):
    out = []
    for h in range(n_hops + 1):
        for s in range(2):
            out.append((TYPES[(s + h) % 2], [2 * (h + 1) + s] if h < n_hops else []))
    return out


def sample_nodes_faithful(graph, num_samples, head_links, rs):
    """``generator.py:298-333``: the six (type, id list) pairs of one batch."""
    schema = sampling_schema(num_samples)
    nodes_by_type = []
    for ii in range(2):
        head_nodes = [e[ii] for e in head_links]
        node_samples = _walker.walk_sequential(graph, num_samples, head_nodes, rs)
        nodes_by_type.append([
            (nt, reduce(operator.concat,
                        (samples[ks] for samples in node_samples for ks in indices), []))
            for nt, indices in schema[ii]
        ])
    return [
        (ab[0][0], reduce(operator.concat, (ab[0][1], ab[1][1])))
        for ab in zip(nodes_by_type[0], nodes_by_type[1])
    ]


def get_features(graph, nodes_by_type, head_size):
    """``generator.py:270-277``."""
    feats = [graph.get_node_features_from_node(layer_nodes, node_type)
             for node_type, layer_nodes in nodes_by_type]
    return [np.reshape(a, (head_size, -1, a.shape[1])) for a in feats]


def sample_features_faithful(graph, num_samples, head_links, rs):
    nodes_by_type = sample_nodes_faithful(graph, num_samples, head_links, rs)
    return get_features(graph, nodes_by_type, len(head_links))


def slots_from_hops(head_links, hops):
    """Slot id arrays from dense per-hop arrays ``[2B, size_h]`` whose first B
    rows belong to the user heads and last B rows to the item heads."""
    links = np.asarray(head_links, dtype=np.int64).reshape(-1, 2)
    b = links.shape[0]
    slots = [links[:, 0].copy(), links[:, 1].copy()]
    for hop in hops:
        slots.append(np.asarray(hop[:b]).reshape(-1))
        slots.append(np.asarray(hop[b:]).reshape(-1))
    return slots


def sample_features_fast(graph_features, num_users, head_links, hops):
    """x0..x(2L+1) from dense hop arrays; features as float arrays per type."""
    b = len(head_links)
    out = []
    for s, ids in enumerate(slots_from_hops(head_links, hops)):
        node_type = TYPES[(s % 2 + s // 2) % 2]
        rows = ids if node_type == 'user' else ids - num_users
        a = graph_features[node_type][rows, :]
        out.append(a.reshape(b, -1, a.shape[1]))
    return out


class OracleLinkSequence:
    def __init__(self, sample_function, batch_size, ids, targets=None, shuffle=True, seed=None):
        self.targets = None if targets is None else np.asanyarray(targets)
        self._sample_features = sample_function
        self.batch_size = batch_size
        self.ids = list(ids)
        self.data_size = len(self.ids)
        self.shuffle = shuffle
        self._rs = _stdlib_random if seed is None else _stdlib_random.Random(seed)
        self.on_epoch_end()

    def __len__(self):
        return int(np.ceil(self.data_size / self.batch_size))

    def __getitem__(self, batch_num):
        start = self.batch_size * batch_num
        end = start + self.batch_size
        if start >= self.data_size:
            raise IndexError("Mapper: batch_num larger than length of data")
        batch_indices = self.indices[start:end]
        head_ids = [self.ids[ii] for ii in batch_indices]
        batch_targets = self.targets[batch_indices]
        return self._sample_features(head_ids, batch_num), batch_targets

    def on_epoch_end(self):
        self.indices = list(range(self.data_size))
        if self.shuffle:
            self._rs.shuffle(self.indices)
