"""Seed plumbing and the mean aggregator layer.

``random_state`` / ``is_real_iterable`` keep the behaviour of ``utils.py:15-33``.

``Aggregator`` keeps the constructor and call convention of the reference layer
(``utils.py:36-120``): ``Aggregator(output_dim, has_bias, activation_function,
agg_option)`` applied to ``[x_self (B, heads, Fs), x_neigh (B, heads, n, Fn)]``
gives ``act(concat(x_self @ w_self, mean_n(x_neigh) @ w_neigh) + bias)`` of shape
``(B, heads, output_dim)``; weights are created on first use with the shapes and
initialisers of ``build`` (``:64-91``: glorot-uniform ``(F, output_dim/2)``
matrices, zero bias).  The arithmetic runs in two HIP kernels: the neighbor mean
(``gs_gather_mean_fwd``) and the MFMA projection (``gs_sage_project_fwd``).
"""
import collections.abc
import math
import random as rn
from collections import namedtuple

import numpy.random as np_rn

from . import autograd as _ag

RandomState = namedtuple("RandomState", "random, numpy")

_rs = RandomState(rn, np_rn)


def random_state(seed):
    """(stdlib, numpy) generators: the process-wide ones for ``None``, freshly
    seeded ones otherwise."""
    if seed is None:
        return _rs
    return RandomState(rn.Random(seed), np_rn.RandomState(seed))


def is_real_iterable(x):
    return isinstance(x, collections.abc.Iterable) and not isinstance(x, (str, bytes))


_ACTIVATIONS = ('relu', 'linear')


class Aggregator:
    def __init__(self, output_dim=32, has_bias=True, activation_function="relu", agg_option='mean',
                 seed=None, **kwargs):
        self.output_dim = output_dim
        if output_dim % 2 != 0:
            raise ValueError("The output_dim must be a multiple of two.")
        self.half_output_dim = output_dim // 2
        self.has_bias = has_bias
        if activation_function is None:
            activation_function = 'linear'
        if activation_function not in _ACTIVATIONS:
            raise ValueError("activation_function must be one of %s (the two the model uses)" % (_ACTIVATIONS,))
        self.activation = activation_function
        assert agg_option in ['mean', 'max'], "agg_option must be mean or max"
        self.agg_option = agg_option
        self.w_neigh = None
        self.w_self = None
        self.bias = None
        self._seed = seed
        self.built = False

    def __repr__(self):
        return "Aggregator with output_dim {}".format(self.output_dim)

    def build(self, input_shape, device=None):
        """input_shape = [(B, heads, Fs), (B, heads, n, Fn)] as Keras passes it."""
        import torch
        from ._lib import require_gpu
        device = device or require_gpu()
        fs, fn = int(input_shape[0][2]), int(input_shape[1][3])
        gen = torch.Generator(device='cpu')
        if self._seed is None:
            gen.seed()
        else:
            gen.manual_seed(int(self._seed))

        def glorot(fan_in, fan_out):
            limit = math.sqrt(6.0 / (fan_in + fan_out))
            w = (torch.rand(fan_in, fan_out, generator=gen, dtype=torch.float64) * 2 - 1) * limit
            return w.to(torch.float32).to(device).requires_grad_(True)

        # creation order as in build(): w_neigh, then w_self, then bias
        self.w_neigh = glorot(fn, self.half_output_dim)
        self.w_self = glorot(fs, self.half_output_dim)
        if self.has_bias:
            self.bias = torch.zeros(self.output_dim, dtype=torch.float32, device=device, requires_grad=True)
        self.built = True

    def parameters(self):
        return [p for p in (self.w_self, self.w_neigh, self.bias) if p is not None]

    def project(self, x_self2d, x_mean2d):
        """The projection alone, on (M, Fs) and (M, Fn) row matrices."""
        return _ag.SageProject.apply(x_self2d, x_mean2d, self.w_self, self.w_neigh, self.bias, self.activation)

    def __call__(self, x, **kwargs):
        x_self, x_neigh = x[0], x[1]
        if not self.built:
            self.build([tuple(x_self.shape), tuple(x_neigh.shape)], device=x_self.device)
        b, heads, fs = x_self.shape
        n, fn = x_neigh.shape[2], x_neigh.shape[3]
        m = b * heads
        reduce = _ag.GroupMax if self.agg_option == 'max' else _ag.GroupMean
        mean = reduce.apply(x_neigh.reshape(m, n, fn).contiguous(), m, n)
        out = self.project(x_self.reshape(m, fs).contiguous(), mean)
        return out.reshape(b, heads, self.output_dim)
