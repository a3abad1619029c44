"""CPU restatement of the reference model arithmetic -- test infrastructure.

PARITY UNPINNED: the reference computes this part with TensorFlow
(``utils.py:109-119``, ``models.py:32,152-168,181-190``, ``link_utils.py:167-168,
244-247``); TensorFlow is not installed here and the reference holds no tests or
stored outputs for it, so this file restates the cited lines and is checked only
against itself in float64.

Written with torch CPU tensors so that the same statement gives gradients by
autograd; ``dtype=torch.float64`` is the arbiter for the fp32 tolerance checks.

* ``aggregator``: ``utils.py:103-120`` -- mean over the neighbor axis, then the
  two matmuls, concat, bias, activation.
* ``glorot_uniform``: the initialiser named at ``utils.py:75,81`` (Keras:
  U(-l, l), l = sqrt(6 / (fan_in + fan_out))); bias zeros (``utils.py:88``).
* ``LayerPlan``: ``models.py:48-90`` -- which slots are alive in which layer,
  their depth, dimensions, activations, and one weight set per (layer, type).
* ``forward``: ``models.py:134-190`` -- reshape the neighbor slot to
  ``(B, heads, n_samples[depth], dim)``, aggregate, after the last layer drop the
  unit axis and l2-normalise (TF semantics: ``x * rsqrt(max(sum(x*x), 1e-12))``).
* ``link_head``: ``link_utils.py:167-168`` (concat) + ``:244`` Dense(1, sigmoid)
  + ``:247`` reshape to ``(B, 1)``.
"""
import math

import numpy as np
import torch

from .generator import type_adjacency_list


def glorot_uniform(fan_in, fan_out, gen, dtype=torch.float32):
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(fan_in, fan_out, generator=gen, dtype=torch.float64) * 2 - 1) * limit).to(dtype)


def aggregator(x_self, x_neigh, w_self, w_neigh, bias, act, agg='mean'):
    """x_self (B,h,Fs), x_neigh (B,h,n,Fn) -> (B,h,2*half).  ``agg`` 'mean' or
    'max' (``utils.py:108-111``); ``amax`` shares its gradient evenly between
    tied maxima, as TensorFlow's reduce_max does."""
    reduced = x_neigh.amax(dim=2) if agg == 'max' else x_neigh.mean(dim=2)
    neigh_agg = torch.matmul(reduced, w_neigh)
    self_agg = torch.matmul(x_self, w_self)
    out = torch.cat([self_agg, neigh_agg], dim=2)
    if bias is not None:
        out = out + bias
    if act == 'relu':
        out = torch.relu(out)
    elif act != 'linear':
        raise ValueError(act)
    return out


def l2_normalize(x, eps=1e-12):
    sq = (x * x).sum(dim=-1, keepdim=True)
    return x * torch.rsqrt(torch.clamp(sq, min=eps))


class LayerPlan:
    """Quantities ``GraphSAGE.__init__`` derives from the generator."""

    def __init__(self, layer_sizes, n_samples, input_dims):
        self.n_samples = list(n_samples)
        self.n_layers = len(self.n_samples)
        self.input_dims = dict(input_dims)
        self.schema = type_adjacency_list(self.n_layers)
        self.neigh_trees = self._trees([li for li in self.schema if len(li[1]) > 0])
        stack = [self.schema] + self.neigh_trees
        self.depths = [self.n_layers + 1 - sum(1 for li in stack if i < len(li))
                       for i in range(len(self.schema))]
        self.dims = []
        for layer, dim in enumerate([self.input_dims] + list(layer_sizes)):
            self.dims.append(dim if isinstance(dim, dict) else {k: dim for k, _ in stack[layer]})
        self.activations = ['relu'] * (self.n_layers - 1) + ['linear']

    @staticmethod
    def _trees(tree):
        kept = [li for li in tree if all(j < len(tree) for j in li[1])]
        return [tree] if len(kept) == 0 else [tree] + LayerPlan._trees(kept)

    def weight_shapes(self):
        """{(layer, node_type): (Fs, Fn, out_dim)}: ``utils.py:72-89`` with the
        input shapes ``models.py:151-168`` feeds each aggregator."""
        shapes = {}
        for layer in range(self.n_layers):
            for i, (node_type, neigh) in enumerate(self.neigh_trees[layer]):
                n_type = self.schema[neigh[0]][0]
                shapes[(layer, node_type)] = (self.dims[layer][node_type], self.dims[layer][n_type],
                                              self.dims[layer + 1][node_type])
        return shapes


def init_weights(plan, gen, bias=True, dtype=torch.float32):
    weights = {}
    for key, (fs, fn, out) in sorted(plan.weight_shapes().items()):
        if out % 2 != 0:
            raise ValueError("The output_dim must be a multiple of two.")
        weights[key] = {
            'w_neigh': glorot_uniform(fn, out // 2, gen, dtype),
            'w_self': glorot_uniform(fs, out // 2, gen, dtype),
            'bias': torch.zeros(out, dtype=dtype) if bias else None,
        }
    return weights


def forward(plan, weights, xin, normalize='l2', keep_layers=None, agg='mean', dropout=None):
    """``dropout``: None, or ``hook(layer, slot, role, x)`` returning x behind the Dropout layer
    ``models.py:152`` (role 'neigh') / ``models.py:165`` (role 'head') puts in front of aggregator call ``slot``
    of ``layer`` (see ``oracle.dropout``)."""
    h = list(xin)
    for layer in range(plan.n_layers):
        out = []
        for i, (node_type, neigh) in enumerate(plan.neigh_trees[layer]):
            heads = h[i].shape[1]
            j = neigh[0]
            n_dim = plan.dims[layer][plan.schema[j][0]]
            x_neigh = h[j].reshape(h[j].shape[0], heads, plan.n_samples[plan.depths[i]], n_dim)
            w = weights[(layer, node_type)]
            x_head = h[i]
            if dropout is not None:
                x_neigh = dropout(layer, i, 'neigh', x_neigh)
                x_head = dropout(layer, i, 'head', x_head)
            out.append(aggregator(x_head, x_neigh, w['w_self'], w['w_neigh'], w['bias'],
                                  plan.activations[layer], agg))
        h = out
        if keep_layers is not None:
            keep_layers.append(h)
    h = [x.reshape(x.shape[0], x.shape[2]) for x in h if x.shape[1] == 1]
    if normalize == 'l2':
        h = [l2_normalize(x) for x in h]
    return h[0] if len(h) == 1 else h


def link_head(user_emb, item_emb, kernel, bias):
    """concat -> Dense(1, sigmoid) -> (B, 1)."""
    z = torch.cat([user_emb, item_emb], dim=-1) @ kernel + bias
    return torch.sigmoid(z).reshape(-1, 1)


def mse_loss(pred, target):
    """Keras ``mean_squared_error`` averaged over the batch."""
    return ((pred - target.reshape(pred.shape)) ** 2).mean()


def keras_auc(y_true, y_pred, num_thresholds=200):
    """Keras ``metrics.AUC()`` defaults: ROC curve, 200 thresholds, Riemann sum by
    interpolation (trapezoid over the FPR axis)."""
    y_true = np.asarray(y_true).reshape(-1).astype(bool)
    y_pred = np.asarray(y_pred).reshape(-1).astype(np.float64)
    eps = 1e-7
    th = [0.0 - eps] + [(i + 1) / (num_thresholds - 1) for i in range(num_thresholds - 2)] + [1.0 + eps]
    th = np.asarray(th)[:, None]
    pred_pos = y_pred[None, :] > th
    tp = (pred_pos & y_true[None, :]).sum(1).astype(np.float64)
    fp = (pred_pos & ~y_true[None, :]).sum(1).astype(np.float64)
    fn = (~pred_pos & y_true[None, :]).sum(1).astype(np.float64)
    tn = (~pred_pos & ~y_true[None, :]).sum(1).astype(np.float64)
    tpr = tp / np.maximum(tp + fn, 1e-7) if (tp + fn).max() > 0 else tp * 0
    fpr = fp / np.maximum(fp + tn, 1e-7) if (fp + tn).max() > 0 else fp * 0
    return float(np.sum((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2.0))
