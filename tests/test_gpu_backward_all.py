"""gs_sage2_backward_all on random saved tables against the same sums written out in float64 (an
independent restatement of models.py:32, utils.py:103-119 backward and the weight gradients of both
layers; train_movielens.py:183-189 for the head form is covered by tests/test_gpu_train.py)."""
import numpy as np
import pytest
import torch

from graphsage_amd import kernels

pytestmark = pytest.mark.gpu
TYPES = ('user', 'item')


def _case(b, n0, f, normalize, act, masked, by_id, scale, seed, dev):
    rng = np.random.default_rng(seed)
    h1w, h2 = 32, 32
    rows = b * (1 + n0)
    nd = lambda *s: rng.standard_normal(s).astype(np.float32)
    t = {}
    t['z'] = {k: nd(b, h2) for k in TYPES}
    t['g_emb'] = {k: nd(b, h2) for k in TYPES}
    t['w1'] = {k: (nd(h1w, h2 // 2), nd(h1w, h2 // 2)) for k in TYPES}
    t['h1'] = {k: np.maximum(nd(b, h1w), 0) for k in TYPES}                    # relu outputs: zeros where inactive
    t['x_mean1'] = {k: nd(b, h1w) for k in TYPES}
    t['x_mean'] = {k: nd(rows, f) for k in TYPES}
    t['mask'] = {k: rng.integers(0, 2 ** 32, b * n0, dtype=np.uint64).astype(np.uint32) for k in TYPES}
    if by_id:
        n_table, first = 300, 1000
        t['table'] = {k: nd(n_table, f) for k in TYPES}
        ids = {k: rng.integers(first, first + n_table, rows).astype(np.int32) for k in TYPES}
        for k in TYPES:                                                        # a few ids outside the table: rows of zeros
            ids[k][rng.integers(0, rows, 5)] = first - 1
            ids[k][rng.integers(0, rows, 5)] = first + n_table
        t['ids'], t['first'] = ids, first
        t['x_self'] = {}
        for k in TYPES:
            r = ids[k].astype(np.int64) - first
            ok = (r >= 0) & (r < n_table)
            xs = t['table'][k][np.where(ok, r, 0)]
            xs[~ok] = 0
            t['x_self'][k] = xs
    else:
        t['x_self'] = {k: nd(rows, f) for k in TYPES}
    return t


def _expected(t, b, n0, normalize, act, masked, scale):
    """float64; returns {(layer, type): (g_w_self, g_w_neigh, g_bias)}"""
    out = {}
    g_root, g_heads = {}, {}
    for k, o in (('user', 'item'), ('item', 'user')):
        z, g = t['z'][k].astype(np.float64), t['g_emb'][k].astype(np.float64)
        if normalize:
            ss = (z * z).sum(1, keepdims=True)
            s = 1 / np.sqrt(np.maximum(ss, 1e-12))
            gz = s * g - np.where(ss > 1e-12, s ** 3 * (z * g).sum(1, keepdims=True), 0) * z
        else:
            gz = g
        if act == 'relu':
            gz = gz * (z > 0)
        ws, wn = (w.astype(np.float64) for w in t['w1'][k])
        h1, xm1 = t['h1'][k].astype(np.float64), t['x_mean1'][k].astype(np.float64)
        out[(1, k)] = (h1.T @ gz[:, :16], xm1.T @ gz[:, 16:], gz.sum(0))
        ds = gz[:, :16] @ ws.T * scale
        dn = gz[:, 16:] @ wn.T * scale / n0
        if masked:
            ds = ds * (h1 > 0)
        g_root[k] = ds
        gh = np.repeat(dn, n0, axis=0)                                          # row p*n0 + j of the OTHER type's table
        if masked:
            bits = (t['mask'][o][:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1
            gh = gh * bits
        g_heads[o] = gh
    for k in TYPES:
        g0 = np.concatenate([g_root[k], g_heads[k]], 0)
        xs, xn = t['x_self'][k].astype(np.float64), t['x_mean'][k].astype(np.float64)
        out[(0, k)] = (xs.T @ g0[:, :16], xn.T @ g0[:, 16:], g0.sum(0))
    return out


@pytest.mark.parametrize('b,n0,f,normalize,act,masked,by_id,scale', [
    (37, 25, 64, True, 'linear', True, True, 1.0),          # the benchmark's shape
    (5, 1, 32, True, 'linear', True, False, 1.0),
    (300, 13, 64, False, 'relu', True, True, 1.25),         # four steps per round (n0 <= 16)
    (64, 16, 32, True, 'relu', False, False, 1.0),          # layer 0 without activation
    (41, 17, 32, True, 'linear', True, True, 1.0),          # eight steps per round
    (19, 40, 64, True, 'linear', True, True, 2.0),          # two rounds of eight
    (9, 64, 32, False, 'linear', True, False, 1.0),         # the largest fan-out: a lane per hop-1 head
    (2600, 5, 64, True, 'linear', True, True, 1.0),         # more walks than waves in the grid: several per wave, two passes over the roots
])
def test_whole_backward_equals_float64_sums(b, n0, f, normalize, act, masked, by_id, scale):
    dev = torch.device('cuda', 0)
    t = _case(b, n0, f, normalize, act, masked, by_id, scale, 1234 + b + n0, dev)
    want = _expected(t, b, n0, normalize, act, masked, scale)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    saved = {'z': {k: dv(t['z'][k]) for k in TYPES}, 'h1': {k: dv(t['h1'][k]) for k in TYPES},
             'x_mean1': {k: dv(t['x_mean1'][k]) for k in TYPES}, 'x_mean': {k: dv(t['x_mean'][k]) for k in TYPES},
             'h1_mask': {k: dv(t['mask'][k].view(np.int32)) for k in TYPES} if masked else None}
    if by_id:
        saved['x_self'] = {k: (dv(t['table'][k]), t['first']) for k in TYPES}
        saved['self_ids'] = {k: dv(t['ids'][k]) for k in TYPES}
    else:
        saved['x_self'] = {k: dv(t['x_self'][k]) for k in TYPES}
    weights1 = {k: (dv(t['w1'][k][0]), dv(t['w1'][k][1])) for k in TYPES}
    grads = {}
    for layer, kdim in ((0, f), (1, 32)):
        for k in TYPES:
            grads[(layer, k)] = (torch.full((kdim, 16), 7.0, device=dev), torch.full((kdim, 16), 7.0, device=dev),
                                 torch.full((32,), 7.0, device=dev))
    call = kernels.Sage2BackwardAll(saved, weights1, grads, n0, normalize, act, head=None, input_scale=scale)
    g_emb = {k: dv(t['g_emb'][k]) for k in TYPES}
    call.run(g_emb=g_emb)
    first = {key: [x.clone() for x in val] for key, val in grads.items()}
    for key, (ws, wn, bias) in want.items():
        for name, got, ref in zip(('w_self', 'w_neigh', 'bias'), grads[key], (ws, wn, bias)):
            got = got.cpu().numpy().astype(np.float64).reshape(ref.shape)
            tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
            assert np.abs(got - ref).max() <= tol, (key, name, float(np.abs(got - ref).max()), tol)
    call.run(g_emb=g_emb)                                   # the same bits again (fixed order of every sum)
    for key in grads:
        for x, y in zip(grads[key], first[key]):
            assert torch.equal(x, y)


def test_shapes_outside_the_kernel_are_refused():
    from graphsage_amd._lib import GsageError
    assert kernels.sage2_backward_all_supported(64, 64, 32, 32) and kernels.sage2_backward_all_supported(32, 32, 32, 32)
    assert not kernels.sage2_backward_all_supported(64, 64, 64, 64)
    assert not kernels.sage2_backward_all_supported(64, 32, 32, 32)
    dev = torch.device('cuda', 0)
    t = _case(4, 65, 32, True, 'linear', True, False, 1.0, 1, dev)              # fan-out beyond a lane per head
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    saved = {'z': {k: dv(t['z'][k]) for k in TYPES}, 'h1': {k: dv(t['h1'][k]) for k in TYPES},
             'x_mean1': {k: dv(t['x_mean1'][k]) for k in TYPES}, 'x_mean': {k: dv(t['x_mean'][k]) for k in TYPES},
             'h1_mask': {k: dv(t['mask'][k].view(np.int32)) for k in TYPES}, 'x_self': {k: dv(t['x_self'][k]) for k in TYPES}}
    grads = {(layer, k): (torch.zeros(32, 16, device=dev), torch.zeros(32, 16, device=dev), None) for layer in (0, 1) for k in TYPES}
    call = kernels.Sage2BackwardAll(saved, {k: (dv(t['w1'][k][0]), dv(t['w1'][k][1])) for k in TYPES}, grads, 65, True, 'linear')
    with pytest.raises(GsageError, match='at most 64'):
        call.run(g_emb={k: dv(t['g_emb'][k]) for k in TYPES})
