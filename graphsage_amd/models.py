"""Two-type GraphSAGE layer stack for user-item link batches.

Constructor and call convention of the reference's ``GraphSAGE``
(``models.py:14-245``): ``GraphSAGE(layer_sizes, generator, agg_option, bias,
dropout, normalize)``, ``model(xin)`` on the batch the generator produced gives
``[user_embedding, item_embedding]``, each ``(B, layer_sizes[-1])``.

Layer wiring (``models.py:48-90,134-190``), stated directly instead of through
the reference's generic tree walk: with L hops the batch has slots
``s = 2*h + side`` (``h`` = hop, ``side`` = which head the walk started from);
layer ``l`` keeps slots ``0 .. 2*(L-l)-1``; slot ``s`` aggregates itself (head)
with slot ``s + 2`` (its sampled neighbors, ``n_samples[h]`` per head node); one
weight set exists per (layer, node type of the head slot) and is shared by all
slots of that type in the layer; activations are relu except linear on the last
layer; the surviving two slots are reshaped to ``(B, H)`` and l2-normalised.

Two execution paths produce the same numbers:

* dense -- ``xin`` is a list of ``(B, n, F)`` tensors (the reference's input
  format): every layer is neighbor mean + projection on materialised tensors;
* fused -- ``xin`` is the generator's ``BatchFeatures``: layer 0 reads the
  deepest slots straight from the feature tables through their sampled ids
  (``gs_gather_mean_fwd``), so the largest tensors of the batch, ``(B, n0*n1, F)``,
  are never written or read back.
"""
import itertools as it
import operator as op

from . import autograd as _ag
from .utils import Aggregator

__all__ = ['GraphSAGE']


class GraphSAGE:
    def __init__(self, layer_sizes, generator=None, agg_option='mean', bias=True, dropout=0.0,
                 normalize="l2", seed=None):
        self._aggregator = Aggregator
        self.agg_option = agg_option
        self.normalize = normalize
        if generator is not None:
            self._get_sizes_from_generator(generator)
        else:
            raise ValueError("Generator must be inserted")
        self.n_layers = len(self.n_samples)
        if len(layer_sizes) != self.n_layers:
            raise ValueError("one layer size per hop is needed (%d hops, %d sizes)" %
                             (self.n_layers, len(layer_sizes)))
        self.bias = bias
        self.dropout = dropout
        self.training = True
        n_slots = len(self.subtree_schema)

        # slots alive at the input of each layer, with their neighbor slot
        self.neigh_trees = [self.subtree_schema[:2 * (self.n_layers - layer)] for layer in range(self.n_layers)]
        self._depths = [s // 2 for s in range(n_slots)]
        types = sorted({t for t, _ in self.subtree_schema}, reverse=True)      # ['user', 'item']
        self.dims = [dict(self.input_dims)] + [{t: int(size) for t in types} for size in layer_sizes]
        self.activations = ["relu"] * (self.n_layers - 1) + ["linear"]
        self._aggs = [
            {
                node_type: self._aggregator(
                    output_dim, has_bias=self.bias, activation_function=self.activations[layer],
                    agg_option=self.agg_option,
                    seed=None if seed is None else seed * 1000 + layer * 10 + ti)
                for ti, (node_type, output_dim) in enumerate(self.dims[layer + 1].items())
            }
            for layer in range(self.n_layers)
        ]

    def __repr__(self):
        return "<GraphSAGE model with {} layers>".format(self.n_layers)

    def _get_sizes_from_generator(self, generator):
        self.input_dims = generator.graph.node_feature_sizes
        self.n_samples = generator.num_samples
        self.subtree_schema = generator.type_adjacency_list

    # ---- parameters ---------------------------------------------------------
    def build(self, device=None):
        """Creates every aggregator's weights (Keras does this on first call)."""
        for layer in range(self.n_layers):
            for node_type, agg in self._aggs[layer].items():
                if agg.built:
                    continue
                other = [t for t in self.dims[layer] if t != node_type][0]
                agg.build([(None, 1, self.dims[layer][node_type]),
                           (None, 1, 1, self.dims[layer][other])], device=device)
        return self

    def parameters(self):
        return [p for layer in self._aggs for agg in layer.values() for p in agg.parameters()]

    def named_parameters(self):
        out = {}
        for layer, aggs in enumerate(self._aggs):
            for node_type, agg in aggs.items():
                out['l%d.%s.w_self' % (layer, node_type)] = agg.w_self
                out['l%d.%s.w_neigh' % (layer, node_type)] = agg.w_neigh
                if agg.bias is not None:
                    out['l%d.%s.bias' % (layer, node_type)] = agg.bias
        return out

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ---- forward ------------------------------------------------------------------
    def _drop(self, x):
        if self.dropout and self.training:
            import torch
            return torch.nn.functional.dropout(x, p=self.dropout, training=True)
        return x

    def _finish(self, h):
        out = [x.reshape(x.shape[0], x.shape[2]) for x in h if x.shape[1] == 1]
        if self.normalize == "l2":
            out = [_ag.L2Normalize.apply(x.contiguous()) for x in out]
        return out[0] if len(out) == 1 else out

    def _dense_layer(self, h, layer):
        out = []
        for i, (node_type, neigh_indices) in enumerate(self.neigh_trees[layer]):
            j = neigh_indices[0]
            heads = h[i].shape[1]
            n = self.n_samples[self._depths[i]]
            f_neigh = self.dims[layer][self.subtree_schema[j][0]]
            x_neigh = self._drop(h[j].reshape(h[j].shape[0], heads, n, f_neigh))
            out.append(self._aggs[layer][node_type]([self._drop(h[i]), x_neigh]))
        return out

    def _fused_first_layer(self, batch):
        alive = self.neigh_trees[0]
        b = batch.head_size
        x_self = []
        for i, (node_type, _) in enumerate(alive):
            feat, off = batch.table(node_type)
            x_self.append(_ag.GatherRows.apply(feat, batch.ids[i].reshape(-1), off))
        out = []
        for i, (node_type, neigh_indices) in enumerate(alive):
            j = neigh_indices[0]
            m = x_self[i].shape[0]
            n = self.n_samples[self._depths[i]]
            group, gather = ((_ag.GroupMax, _ag.GatherMax) if self.agg_option == 'max'
                             else (_ag.GroupMean, _ag.GatherMean))
            if j < len(alive):            # neighbor slot is itself gathered: mean (max) of those rows
                x_mean = group.apply(x_self[j].reshape(m, n, -1), m, n)
            else:                          # deepest slot: rows are reduced as they are fetched
                feat, off = batch.table(self.subtree_schema[j][0])
                x_mean = gather.apply(feat, batch.ids[j].reshape(-1), m, n, off)
            agg = self._aggs[0][node_type]
            out.append(agg.project(x_self[i], x_mean).reshape(b, m // b, agg.output_dim))
        return out

    def __call__(self, xin):
        self.build()
        self.layer_tensors = []
        from .generator import BatchFeatures
        if isinstance(xin, BatchFeatures) and not (self.dropout and self.training):
            h = self._fused_first_layer(xin)
            self.layer_tensors.append(h)
            first = 1
        else:
            h = [xin[i] for i in range(len(self.subtree_schema))]
            first = 0
        for layer in range(first, self.n_layers):
            h = self._dense_layer(h, layer)
            self.layer_tensors.append(h)
        return self._finish(h)

    # ---- shapes -------------------------------------------------------------------
    def _input_shapes(self):
        """(nodes per head, feature size) of every slot."""
        sizes = list(it.accumulate([1] + list(self.n_samples), op.mul))
        return [(sizes[self._depths[s]], self.input_dims[t]) for s, (t, _) in enumerate(self.subtree_schema)]

    def in_out_tensors(self):
        """The reference returns Keras placeholders and the symbolic output
        (``models.py:229-245``); without a graph-building framework the
        equivalent is the list of per-slot input shapes and the model itself as
        the function from a batch to ``[user_embedding, item_embedding]``."""
        return self._input_shapes(), self
