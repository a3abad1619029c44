"""CPU restatement of the reference graph store -- test infrastructure.

Follows ``bipartite_graph.py:9-93``: a node-type dict built from the ``nodes``
lists (``:48-51``), an adjacency dict keyed by the source nodes that occur in
``edges[0]``, each value listing ``edges[1]`` of the matching columns in column
order (``:53-60``), and feature rows addressed by ``id`` for users and
``id - num_users`` for items (``:31,70-76``).

The reference fills the adjacency dict with one ``np.where`` scan per node
(O(N*E)); a stable sort by source yields the same lists, and that is what is
done here so that the oracle stays usable at a few million edges.
"""
import numpy as np


class OracleGraph:
    def __init__(self, node_features, nodes, edges, lazy_csr=None):
        if not isinstance(node_features, dict):
            raise TypeError("node features must be python dictionary")
        self.node_features = node_features
        self.nodes = nodes
        self.edges = edges
        self.node_type_dict = {n: 'user' for n in nodes['user']}
        self.node_type_dict.update({n: 'item' for n in nodes['item']})
        self.correction_constant = node_features['user'].shape[0]
        self._lazy = lazy_csr
        if lazy_csr is None:
            self.neighbor_dict = self._neighbor_dict(np.asarray(edges))
        else:
            self.neighbor_dict = _LazyNeighbors(*lazy_csr)

    @staticmethod
    def _neighbor_dict(edges):
        src = edges[0]
        order = np.argsort(src, kind='stable')
        s_sorted = src[order]
        d_sorted = edges[1][order]
        keys, starts = np.unique(s_sorted, return_index=True)
        ends = np.append(starts[1:], len(s_sorted))
        return {int(k): d_sorted[a:b].tolist() for k, a, b in zip(keys, starts, ends)}

    @classmethod
    def from_csr(cls, node_features, num_users, num_items, indptr, indices):
        """Oracle view over an existing CSR (for timing on graphs too large to
        hold as Python lists): neighbor lists are converted on first touch and
        cached, so a second pass over the same walks sees plain lists exactly as
        the reference's ``neighbor_dict`` holds them."""
        nodes = {'user': range(num_users), 'item': range(num_users, num_users + num_items)}
        g = cls.__new__(cls)
        g.node_features = node_features
        g.nodes = nodes
        g.edges = None
        g.node_type_dict = _RangeTypes(num_users, num_items)
        g.correction_constant = num_users
        g.neighbor_dict = _LazyNeighbors(np.asarray(indptr), np.asarray(indices))
        g._lazy = True
        return g

    def get_neighbors_from_node(self, node):
        return self.neighbor_dict[node]

    def get_node_type(self, node):
        return self.node_type_dict[node]

    def get_node_features_from_node(self, nodes, node_type):
        if node_type == 'user':
            rows = nodes
        else:
            rows = [n - self.correction_constant for n in nodes]
        return self.node_features[node_type][rows, :]

    @property
    def num_nodes(self):
        return len(self.node_type_dict)

    @property
    def num_edges(self):
        return self.edges.shape[1]

    @property
    def node_feature_sizes(self):
        return {'user': self.node_features['user'].shape[1],
                'item': self.node_features['item'].shape[1]}


class _RangeTypes:
    """Stand-in for the node-type dict of a graph whose ids are the two ranges
    [0, NU) and [NU, NU+NI)."""

    def __init__(self, nu, ni):
        self.nu, self.ni = nu, ni

    def __getitem__(self, node):
        if 0 <= node < self.nu:
            return 'user'
        if self.nu <= node < self.nu + self.ni:
            return 'item'
        raise KeyError(node)

    def __len__(self):
        return self.nu + self.ni

    def keys(self):
        return range(self.nu + self.ni)


class _LazyNeighbors:
    def __init__(self, indptr, indices):
        self.indptr, self.indices = indptr, indices
        self.cache = {}

    def __getitem__(self, node):
        hit = self.cache.get(node)
        if hit is None:
            a, b = int(self.indptr[node]), int(self.indptr[node + 1])
            if a == b:
                raise KeyError(node)
            hit = self.indices[a:b].tolist()
            self.cache[node] = hit
        return hit
