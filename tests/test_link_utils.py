"""Edge-embedding operators and heads of link_utils.py (reference link_utils.py:104-256) on CPU tensors."""
import numpy as np
import pytest
import torch

from graphsage_amd.link_utils import EdgeHead, LinkEmbedding, link_inference


def _pair(n=7, m=5, seed=0):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, m)).astype(np.float32), rng.standard_normal((n, m)).astype(np.float32)


@pytest.mark.parametrize('method,fn', [
    ('ip', lambda a, b: (a * b).sum(-1, keepdims=True)), ('dot', lambda a, b: (a * b).sum(-1, keepdims=True)),
    ('l1', lambda a, b: np.abs(a - b)), ('l2', lambda a, b: (a - b) ** 2),
    ('mul', lambda a, b: a * b), ('hadamard', lambda a, b: a * b),
    ('concat', lambda a, b: np.concatenate([a, b], -1)), ('avg', lambda a, b: (a + b) / 2)])
def test_link_embedding_operators(method, fn):
    a, b = _pair()
    le = LinkEmbedding(method=method.upper())                         # the reference lower-cases the name
    got = le([torch.from_numpy(a), torch.from_numpy(b)]).numpy()
    np.testing.assert_allclose(got, fn(a, b), rtol=1e-6, atol=1e-6)
    stacked = torch.from_numpy(np.stack([a, b], axis=1))              # (N, 2, M): the full-batch form
    np.testing.assert_allclose(le(stacked).numpy(), fn(a, b), rtol=1e-6, atol=1e-6)
    assert le.width(5) == fn(a, b).shape[-1]


def test_link_embedding_rejects_what_the_reference_rejects():
    a, b = _pair()
    with pytest.raises(ValueError):
        LinkEmbedding()([torch.from_numpy(a)])
    with pytest.raises(ValueError):
        LinkEmbedding()(torch.zeros(4, 3, 5))
    with pytest.raises(TypeError):
        LinkEmbedding()(a)
    with pytest.raises(NotImplementedError):
        LinkEmbedding(method='xor')([torch.from_numpy(a), torch.from_numpy(b)])
    with pytest.raises(NotImplementedError):
        link_inference(edge_embedding_method='xor')


def test_edge_head_dense_activation_reshape_and_clip():
    a, b = _pair(n=9, m=4, seed=3)
    x = [torch.from_numpy(a), torch.from_numpy(b)]
    head = EdgeHead(4, 'cpu', output_dim=3, output_act='softmax', method='l2', seed=1)
    assert [tuple(p.shape) for p in head.parameters()] == [(4, 3), (3,)]
    limit = np.sqrt(6.0 / (4 + 3))
    assert float(head.kernel.abs().max()) <= limit and float(head.bias.abs().max()) == 0.0
    out = head(x)
    z = ((a - b) ** 2) @ head.kernel.detach().numpy()
    want = np.exp(z - z.max(-1, keepdims=True))
    want /= want.sum(-1, keepdims=True)
    np.testing.assert_allclose(out.detach().numpy(), want, rtol=1e-5, atol=1e-6)
    # inner product: no Dense layer, output_dim forced to 1 (with the reference's warning)
    with pytest.warns(UserWarning):
        link_inference(output_dim=4, edge_embedding_method='ip')
    ip = EdgeHead(4, 'cpu', output_act='sigmoid', method='ip')
    assert ip.parameters() == []
    np.testing.assert_allclose(ip(x).numpy(), 1 / (1 + np.exp(-(a * b).sum(-1, keepdims=True))), rtol=1e-6)
    # LeakyClippedLinear(low, high, alpha=0.1) of link_regression
    reg = EdgeHead(4, 'cpu', method='ip', clip_limits=(-0.5, 0.5))
    v = (a * b).sum(-1, keepdims=True)
    want = np.where(v < -0.5, -0.5 + 0.1 * (v + 0.5), np.where(v > 0.5, 0.5 + 0.1 * (v - 0.5), v))
    np.testing.assert_allclose(reg(x).numpy(), want, rtol=1e-6, atol=1e-7)
    out.sum().backward()
    assert head.kernel.grad is not None
