"""User-item bipartite graph store: CSR on the host, replicated CSR + feature
tables in HBM.

Keeps the interface of the reference's ``Graph`` (``bipartite_graph.py:9-93``):
same constructor arguments, ``get_neighbors_from_node`` / ``get_node_type`` /
``get_node_features_from_node``, ``num_nodes`` / ``num_edges`` /
``node_feature_sizes``, ``in`` and ``repr``.  What differs is the storage: the
reference keeps ``dict[node] -> list`` built with one ``np.where`` scan per node
(``:53-60``); here the directed edge list is stably sorted by source once, which
yields the same per-node neighbor order (edge-column order, duplicates kept) as
``indptr int64[N+1]`` / ``indices int32[E]`` -- the layout the sampler kernel
reads.

Id space (``bipartite_graph.py:31,70-76``): users are ``[0, NU)`` with
``NU = node_features['user'].shape[0]`` and item ``i`` lives at id ``NU + i``;
the item feature row of node ``v`` is ``v - NU``.
"""
import numpy as np

__all__ = ['Graph', 'IsolatedNodeError']


class IsolatedNodeError(KeyError, ValueError):
    """Sampling reached a node with no outgoing edge.

    The reference fails in one of two ways for such a node: ``KeyError`` from
    the adjacency dict when the node never occurs as a source
    (``bipartite_graph.py:63``), ``ValueError`` when its list exists but is
    empty (``sampler.py:81-82``).  A CSR cannot tell the two apart, so this
    error is both.
    """

    def __str__(self):
        return str(self.args[0]) if self.args else ''


class Graph(object):
    def __init__(self, node_features=None, nodes=None, edges=None):
        if not isinstance(node_features, dict):
            raise TypeError("node features must be python dictionary")
        self.node_features = node_features
        self.nodes = nodes
        self.edges = edges
        self.correction_constant = int(node_features['user'].shape[0])
        self._num_users = self.correction_constant
        self._num_items = int(node_features['item'].shape[0])
        n = self._num_users + self._num_items

        # membership and type, the role of node_type_dict (bipartite_graph.py:48-51)
        self._present = np.zeros(n, dtype=bool)
        self._n_declared = 0
        for node_type, lo, hi in (('user', 0, self._num_users), ('item', self._num_users, n)):
            ids = np.asarray(list(nodes[node_type]), dtype=np.int64)
            if ids.size and (ids.min() < lo or ids.max() >= hi):
                raise ValueError(
                    "%s ids must lie in [%d, %d): users are numbered from 0 and items from the "
                    "number of user feature rows" % (node_type, lo, hi))
            self._present[ids] = True
        self._n_declared = int(self._present.sum())

        e = np.asarray(edges)
        if e.ndim != 2 or e.shape[0] != 2:
            raise ValueError("edges must have shape (2, E)")
        if e.size and (e.min() < 0 or e.max() >= n):
            raise ValueError("edge endpoints must be node ids in [0, %d)" % n)
        self.indptr, self.indices = self._build_csr(e, n)
        self._device = None

    @classmethod
    def from_csr(cls, node_features, indptr, indices):
        """Wraps an existing CSR (numpy or torch, host or device) without an edge
        list; every id in range is a node.  Used for graphs generated on the GPU."""
        g = cls.__new__(cls)
        g.node_features = node_features
        g.correction_constant = int(node_features['user'].shape[0])
        g._num_users = g.correction_constant
        g._num_items = int(node_features['item'].shape[0])
        n = g._num_users + g._num_items
        g.nodes = {'user': range(g._num_users), 'item': range(g._num_users, n)}
        g.edges = None
        g._present = None
        g._n_declared = n
        if indptr.shape[0] != n + 1:
            raise ValueError("indptr must have %d entries" % (n + 1))
        g.indptr, g.indices = indptr, indices
        g._device = None
        return g

    @classmethod
    def from_edges_device(cls, node_features, a, b, symmetric=True):
        """Adjacency built on the GPU from edge columns that already live there
        (int32 tensors; ``symmetric``: every edge a[i] -- b[i] is entered in both
        directions, the doubling of ``train_movielens.py:130-132``).  Same CSR as
        the constructor makes from the same edge array; every id in range is a node."""
        from . import kernels
        n = int(node_features['user'].shape[0]) + int(node_features['item'].shape[0])
        indptr, indices = kernels.build_csr(a, b, n, symmetric=symmetric)
        return cls.from_csr(node_features, indptr, indices)

    @staticmethod
    def _build_csr(edges, n):
        src = edges[0].astype(np.int64, copy=False)
        order = np.argsort(src, kind='stable')
        indices = edges[1][order].astype(np.int32)
        indptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(src, minlength=n), out=indptr[1:])
        return indptr, indices

    def __repr__(self):
        return "<Graph Object for User-Item Bipartite Graph with {} nodes>".format(self.num_nodes)

    def _has(self, node):
        try:
            node = int(node)
        except (TypeError, ValueError):
            return False
        if not 0 <= node < self._num_users + self._num_items:
            return False
        return True if self._present is None else bool(self._present[node])

    def __contains__(self, value):
        return self._has(value)

    def get_neighbors_from_node(self, node):
        if not self._has(node):
            raise KeyError(node)
        a, b = int(self.indptr[node]), int(self.indptr[node + 1])
        if a == b:
            raise KeyError(node)
        out = self.indices[a:b]
        return out.tolist()

    def get_node_type(self, node):
        if not self._has(node):
            raise KeyError(node)
        return 'user' if int(node) < self._num_users else 'item'

    def get_node_features_from_node(self, nodes, node_type):
        rows = np.asarray(nodes, dtype=np.int64)
        if node_type != 'user':
            rows = rows - self.correction_constant
        return self.node_features[node_type][rows, :]

    @property
    def num_users(self):
        return self._num_users

    @property
    def num_items(self):
        return self._num_items

    @property
    def num_nodes(self):
        return self._n_declared

    @property
    def num_edges(self):
        if self.edges is not None:
            return self.edges.shape[1]
        return int(self.indices.shape[0])

    @property
    def node_feature_sizes(self):
        return {'user': self.node_features['user'].shape[1],
                'item': self.node_features['item'].shape[1]}

    # ---- cursors of the cyclic walker -----------------------------------------
    def sampling_cursor(self):
        """Per-node cursors of the cyclic walker (``sampler_backup.BreadthFirstWalker``) as a GPU
        tensor int64[num nodes], all zero at first like the reference's ``sampling_index_dict``
        (``bipartite_graph.py:44-46``)."""
        if getattr(self, '_cursor', None) is None:
            import torch
            dev = self.device_tensors()['indptr'].device
            self._cursor = torch.zeros(self._num_users + self._num_items, dtype=torch.int64, device=dev)
        return self._cursor

    @property
    def sampling_index_dict(self):
        """The cursors as the reference's attribute of this name presents them: {node: position in its
        neighbor list} (a snapshot; the walker advances the device copy)."""
        cur = self.sampling_cursor().cpu().tolist()
        return {v: cur[v] for v in range(len(cur)) if self._has(v)}

    # ---- device residency -------------------------------------------------
    def device_tensors(self):
        """CSR and fp32 feature tables as GPU tensors (uploaded once, then kept:
        the graph is replicated per GPU, sampling never communicates)."""
        if self._device is None:
            import torch
            from ._lib import require_gpu
            dev = require_gpu()

            def up(a, dtype):
                if isinstance(a, torch.Tensor):
                    return a.to(device=dev, dtype=dtype).contiguous()
                return torch.as_tensor(np.ascontiguousarray(a), device=dev).to(dtype).contiguous()

            self._device = {
                'indptr': up(self.indptr, torch.int64),
                'indices': up(self.indices, torch.int32),
                'user': up(self.node_features['user'], torch.float32),
                'item': up(self.node_features['item'], torch.float32),
            }
        return self._device
