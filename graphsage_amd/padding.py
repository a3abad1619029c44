"""Zero-padded views of a graph and a model for the kernels that want feature
widths in multiples of 32 (MovieLens has 28 / 18 or 24 / 19 columns).

Zero columns appended to the feature tables and zero rows appended to the
layer-0 weights leave every product unchanged, and the gradient of the real
weight rows is the leading block of the padded gradient.  The views own the
padded copies; ``refresh()`` copies the current real weights in before a forward
pass, ``scatter_grads()`` copies the real rows of the gradients out after a
backward pass (both are plain device copies, so they can sit inside a captured
HIP graph).
"""
import torch

_TYPES = ('user', 'item')


def padded_widths(widths, multiple=32):
    return {t: -(-int(widths[t]) // multiple) * multiple for t in _TYPES}


class PaddedGraphView:
    def __init__(self, graph, widths):
        self._graph = graph
        dev = dict(graph.device_tensors())
        for t in _TYPES:
            src = dev[t]
            if src.shape[1] != widths[t]:
                wide = torch.zeros((src.shape[0], widths[t]), dtype=src.dtype, device=src.device)
                wide[:, :src.shape[1]].copy_(src)
                dev[t] = wide
        self._dev = dev
        self.node_feature_sizes = dict(widths)

    def device_tensors(self):
        return self._dev

    def __getattr__(self, name):                 # everything else is the graph's own
        return getattr(self._graph, name)


class _PaddedAggregator:
    def __init__(self, agg, rows_self, rows_neigh):
        self._agg = agg
        self.bias, self.activation = agg.bias, agg.activation
        dev, hh = agg.w_self.device, agg.w_self.shape[1]
        self.w_self = torch.zeros((rows_self, hh), dtype=torch.float32, device=dev)
        self.w_neigh = torch.zeros((rows_neigh, hh), dtype=torch.float32, device=dev)
        self.w_self.grad = torch.zeros_like(self.w_self)
        self.w_neigh.grad = torch.zeros_like(self.w_neigh)

    def parameters(self):
        return [self.w_self, self.w_neigh] + ([] if self.bias is None else [self.bias])

    def refresh(self):
        with torch.no_grad():
            self.w_self[:self._agg.w_self.shape[0]].copy_(self._agg.w_self)
            self.w_neigh[:self._agg.w_neigh.shape[0]].copy_(self._agg.w_neigh)

    def scatter_grads(self):
        for real, wide in ((self._agg.w_self, self.w_self), (self._agg.w_neigh, self.w_neigh)):
            if real.grad is None:
                real.grad = torch.zeros_like(real)
            real.grad.copy_(wide.grad[:real.shape[0]])


class PaddedModelView:
    """The model as the engines read it (``dims``, ``_aggs``, ``normalize``,
    ``n_samples``), with layer 0 widened to ``widths``; later layers are the
    model's own aggregators."""

    def __init__(self, model, widths):
        model = model.build()
        self._model = model
        other = {'user': 'item', 'item': 'user'}
        self.dims = [dict(widths)] + [dict(d) for d in model.dims[1:]]
        first = {t: _PaddedAggregator(model._aggs[0][t], widths[t], widths[other[t]]) for t in _TYPES}
        self._aggs = [first] + list(model._aggs[1:])
        self.refresh()

    def build(self):
        return self

    def refresh(self):
        for agg in self._aggs[0].values():
            agg.refresh()

    def scatter_grads(self):
        for agg in self._aggs[0].values():
            agg.scatter_grads()

    def __getattr__(self, name):
        return getattr(self._model, name)
