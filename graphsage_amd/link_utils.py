"""Link prediction heads, after the reference's ``link_utils.py``.

``link_classification(edge_embedding_method="concat")`` -- what the training
script uses (``train_movielens.py:179``): concatenate the two node embeddings,
``Dense(1, sigmoid)``, reshape to ``(B, 1)`` (``link_utils.py:167-168,244-247,
312-317``) -- is the head with kernels of its own (``train.LinkHead``: fused
forward + MSE + gradients, part of the static training plan).

The other operators of ``LinkEmbedding.call`` (``link_utils.py:154-176``: ip/dot,
mul/hadamard, l1, l2, avg), other output widths and activations, and the clipped
regression head (``link_regression``, ``LeakyClippedLinear``) are plain torch
modules here; a model with one of them trains through autograd.
"""
import math
import warnings

import torch

from .train import LinkHead

__all__ = ['LinkEmbedding', 'EdgeHead', 'link_inference', 'link_classification', 'link_regression']

_ACTIVATIONS = {
    'linear': lambda x: x, None: lambda x: x,
    'sigmoid': torch.sigmoid, 'relu': torch.relu, 'tanh': torch.tanh,
    'softmax': lambda x: torch.softmax(x, dim=-1),
}


def _activation(name):
    try:
        return _ACTIVATIONS[name]
    except KeyError:
        raise ValueError("unknown activation %r" % (name,))


class LinkEmbedding:
    """Combines the two node embeddings of a link (``link_utils.py:104-180``):
    a list of two (N, M) tensors, or one tensor with a dimension of size 2 at
    ``axis``."""

    def __init__(self, method="ip", axis=-2, activation="linear"):
        self.method = method.lower()
        self.axis = axis
        self.activation = _activation(activation)

    def __call__(self, x):
        if isinstance(x, (list, tuple)):
            if len(x) != 2:
                raise ValueError("Expecting a list of length 2 for link embedding")
            x0, x1 = x
        elif isinstance(x, torch.Tensor):
            if int(x.shape[self.axis]) != 2:
                raise ValueError("Expecting a tensor of shape 2 along specified axis for link embedding")
            x0, x1 = torch.unbind(x, dim=self.axis)
        else:
            raise TypeError("Expected a list, tuple, or Tensor as input")
        if self.method in ("ip", "dot"):
            out = (x0 * x1).sum(dim=-1, keepdim=True)
        elif self.method == "l1":
            out = (x0 - x1).abs()
        elif self.method == "l2":
            out = (x0 - x1) ** 2
        elif self.method in ("mul", "hadamard"):
            out = x0 * x1
        elif self.method == "concat":
            out = torch.cat([x0, x1], dim=-1)
        elif self.method == "avg":
            out = (x0 + x1) / 2
        else:
            raise NotImplementedError
        return self.activation(out)

    def width(self, emb_dim):
        return {'ip': 1, 'dot': 1, 'concat': 2 * emb_dim}.get(self.method, emb_dim)


class EdgeHead:
    """``link_inference``'s edge function (``link_utils.py:236-256``): link
    embedding, then ``Dense(output_dim, output_act)`` (inner product: only the
    activation), reshape to ``(B, output_dim)``, optionally the leaky clip."""

    def __init__(self, emb_dim, device, output_dim=1, output_act="linear", method="ip", clip_limits=None, seed=None):
        self.embed = LinkEmbedding(method=method, activation="linear")
        self.output_dim = output_dim
        self.act = _activation(output_act)
        self.clip_limits = clip_limits
        self.kernel = self.bias = None
        if self.embed.method not in ("ip", "dot"):
            gen = torch.Generator(device='cpu')
            gen.manual_seed(int(seed)) if seed is not None else gen.seed()
            fan_in = self.embed.width(emb_dim)
            limit = math.sqrt(6.0 / (fan_in + output_dim))                    # Keras glorot_uniform
            k = (torch.rand(fan_in, output_dim, generator=gen, dtype=torch.float64) * 2 - 1) * limit
            self.kernel = k.to(torch.float32).to(device).requires_grad_(True)
            self.bias = torch.zeros(output_dim, dtype=torch.float32, device=device, requires_grad=True)

    def parameters(self):
        return [] if self.kernel is None else [self.kernel, self.bias]

    def __call__(self, x_out):
        le = self.embed(x_out)
        out = self.act(le) if self.kernel is None else self.act(le @ self.kernel + self.bias)
        out = out.reshape(-1, self.output_dim)
        if self.clip_limits:
            low, high = self.clip_limits                                       # LeakyClippedLinear, alpha = 0.1
            out = torch.where(out < low, low + 0.1 * (out - low), torch.where(out > high, high + 0.1 * (out - high), out))
        return out


def link_inference(output_dim=1, output_act="linear", edge_embedding_method="ip", clip_limits=None,
                   name="link_inference"):
    if edge_embedding_method in ("ip", "dot") and output_dim != 1:
        warnings.warn("For inner product link method the output_dim will be ignored as it is fixed to be 1.", stacklevel=2)
        output_dim = 1
    LinkEmbedding(method=edge_embedding_method)([torch.zeros(1, 1), torch.zeros(1, 1)])   # unknown methods raise here
    _activation(output_act)

    def edge_function(x_out, seed=None):
        """``x_out`` is what ``GraphSAGE.in_out_tensors()`` returns second (the
        built model); the head takes its embedding width and device from it."""
        model = x_out.build()
        emb_dim, device = model.dims[-1]['user'], model.parameters()[0].device
        if edge_embedding_method == "concat" and output_dim == 1 and output_act == "sigmoid" and not clip_limits:
            return LinkHead(emb_dim, device, seed=seed)                       # the head with its own kernels
        return EdgeHead(emb_dim, device, output_dim, output_act, edge_embedding_method, clip_limits, seed=seed)

    print("{}: using '{}' method to combine node embeddings into edge embeddings".format(name, edge_embedding_method))
    return edge_function


def link_classification(output_dim=1, output_act="sigmoid", edge_embedding_method="ip"):
    return link_inference(output_dim=output_dim, output_act=output_act,
                          edge_embedding_method=edge_embedding_method, name="link_classification")


def link_regression(output_dim=1, clip_limits=None, edge_embedding_method="ip"):
    return link_inference(output_dim=output_dim, output_act="linear", edge_embedding_method=edge_embedding_method,
                          clip_limits=clip_limits, name="link_regression")
