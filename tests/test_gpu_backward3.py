"""gs_sage3_backward_all on random saved tables against the same sums written out in float64 (an independent
restatement of the three-layer backward pass: models.py:32, utils.py:103-119, the weight gradients of all layers)."""
import numpy as np
import pytest
import torch

from graphsage_amd import kernels

pytestmark = pytest.mark.gpu
TYPES = ('user', 'item')
OTHER = {'user': 'item', 'item': 'user'}


def _tables(b, n0, n1, f, seed):
    rng = np.random.default_rng(seed)
    rows0, rows1 = b * (1 + n0 + n0 * n1), b * (1 + n0)
    nd = lambda *s: rng.standard_normal(s).astype(np.float32)
    relu = lambda x: np.maximum(x, 0)
    t = {'z': {}, 'g_emb': {}, 'w1': {}, 'w2': {}, 'h0': {}, 'h1': {}, 'x_mean0': {}, 'x_mean1': {}, 'x_mean2': {},
         'table': {}, 'row_ids': {}, 'x_self': {}}
    n_table, first = 500, 77
    for k in TYPES:
        t['z'][k], t['g_emb'][k] = nd(b, 32), nd(b, 32)
        t['w1'][k], t['w2'][k] = (nd(32, 16), nd(32, 16)), (nd(32, 16), nd(32, 16))
        t['h0'][k], t['h1'][k] = relu(nd(rows0, 32)), relu(nd(rows1, 32))
        t['x_mean0'][k], t['x_mean1'][k], t['x_mean2'][k] = nd(rows0, f), nd(rows1, 32), nd(b, 32)
        t['table'][k] = nd(n_table, f)
        ids = rng.integers(first, first + n_table, rows0).astype(np.int32)
        ids[rng.integers(0, rows0, 4)] = first - 1                  # ids outside the table: rows of zeros
        ids[rng.integers(0, rows0, 4)] = first + n_table
        t['row_ids'][k] = ids
        r = ids.astype(np.int64) - first
        ok = (r >= 0) & (r < n_table)
        xs = t['table'][k][np.where(ok, r, 0)]
        xs[~ok] = 0
        t['x_self'][k] = xs
    t['first'] = first
    return t


def _expected(t, b, n0, n1, normalize, act):
    d = lambda x: x.astype(np.float64)
    out, gp1_seed, a_row, gp1_hop1, gh0 = {}, {}, {}, {}, {}
    for k in TYPES:
        z, g = d(t['z'][k]), d(t['g_emb'][k])
        if normalize:
            ss = (z * z).sum(1, keepdims=True)
            s = 1 / np.sqrt(np.maximum(ss, 1e-12))
            gz = s * g - np.where(ss > 1e-12, s ** 3 * (z * g).sum(1, keepdims=True), 0) * z
        else:
            gz = g
        if act == 'relu':
            gz = gz * (z > 0)
        w2s, w2n = d(t['w2'][k][0]), d(t['w2'][k][1])
        h1s = d(t['h1'][k][:b])
        out[(2, k)] = (h1s.T @ gz[:, :16], d(t['x_mean2'][k]).T @ gz[:, 16:], gz.sum(0))
        gp1_seed[k] = (gz[:, :16] @ w2s.T) * (h1s > 0)
        a_row[k] = gz[:, 16:] @ w2n.T / n0
    for k in TYPES:                                                # hop-1 nodes of type k hang below seeds of the other type
        gp1_hop1[k] = np.repeat(a_row[OTHER[k]], n0, axis=0) * (d(t['h1'][k][b:]) > 0)
    for k in TYPES:
        gp1 = np.concatenate([gp1_seed[k], gp1_hop1[k]], 0)
        out[(1, k)] = (d(t['h0'][k][:b * (1 + n0)]).T @ gp1[:, :16], d(t['x_mean1'][k]).T @ gp1[:, 16:], gp1.sum(0))
    for k in TYPES:
        o = OTHER[k]
        w1s, w1n = d(t['w1'][k][0]), d(t['w1'][k][1])
        seeds = gp1_seed[k][:, :16] @ w1s.T
        b_other = gp1_seed[o][:, 16:] @ d(t['w1'][o][1]).T / n0                  # from the seeds of the other type
        hop1 = np.repeat(b_other, n0, axis=0) + gp1_hop1[k][:, :16] @ w1s.T
        hop2 = np.repeat(gp1_hop1[o][:, 16:] @ d(t['w1'][o][1]).T / n1, n1, axis=0)   # children of the other type's hop-1 nodes
        gh0[k] = np.concatenate([seeds, hop1, hop2], 0) * (d(t['h0'][k]) > 0)
        out[(0, k)] = (d(t['x_self'][k]).T @ gh0[k][:, :16], d(t['x_mean0'][k]).T @ gh0[k][:, 16:], gh0[k].sum(0))
    return out


@pytest.mark.parametrize('b,n0,n1,f,normalize,act', [
    (16, 25, 10, 64, True, 'linear'),       # the benchmark's fan-outs
    (5, 1, 1, 32, True, 'linear'),
    (70, 3, 13, 64, False, 'relu'),
    (9, 7, 17, 32, True, 'linear'),         # eight steps per round
    (3, 2, 32, 64, True, 'linear'),         # the largest second fan-out
    (600, 2, 3, 64, True, 'linear'),        # more seeds than waves of a type: several per wave, two passes over the roots
])
def test_three_hop_backward_equals_float64_sums(b, n0, n1, f, normalize, act):
    dev = torch.device('cuda', 0)
    t = _tables(b, n0, n1, f, 99 + b + n0 + n1)
    want = _expected(t, b, n0, n1, normalize, act)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    saved = {name: {k: dv(t[name][k]) for k in TYPES} for name in ('z', 'h0', 'h1', 'x_mean0', 'x_mean1', 'x_mean2', 'row_ids')}
    saved['feat'] = {k: (dv(t['table'][k]), t['first']) for k in TYPES}
    weights = {1: {k: (dv(t['w1'][k][0]), dv(t['w1'][k][1])) for k in TYPES}, 2: {k: (dv(t['w2'][k][0]), dv(t['w2'][k][1])) for k in TYPES}}
    grads = {}
    for layer, kdim in ((0, f), (1, 32), (2, 32)):
        for k in TYPES:
            grads[(layer, k)] = (torch.full((kdim, 16), 7.0, device=dev), torch.full((kdim, 16), 7.0, device=dev),
                                 torch.full((32,), 7.0, device=dev))
    call = kernels.Sage3BackwardAll(saved, weights, grads, n0, n1, normalize, act)
    g_emb = {k: dv(t['g_emb'][k]) for k in TYPES}
    call.run(g_emb=g_emb)
    first = {key: [x.clone() for x in val] for key, val in grads.items()}
    for key, refs in want.items():
        for name, got, ref in zip(('w_self', 'w_neigh', 'bias'), grads[key], refs):
            got = got.cpu().numpy().astype(np.float64).reshape(ref.shape)
            tol = 3e-5 * max(1.0, float(np.abs(ref).max()))
            assert np.abs(got - ref).max() <= tol, (key, name, float(np.abs(got - ref).max()), tol)
    call.run(g_emb=g_emb)                                       # the same bits again
    for key in grads:
        for x, y in zip(grads[key], first[key]):
            assert torch.equal(x, y)


@pytest.mark.parametrize('b,n0,n1,f,with_mask', [(16, 25, 10, 64, True), (9, 7, 17, 32, False), (70, 3, 13, 64, True)])
def test_rows_below_the_seeds_given_apart_give_the_same_bits(b, n0, n1, f, with_mask):
    """The hop-1 and hop-2 rows as tables of their own (hop-2 outputs as one mask word per row, or as rows) against
    the same rows inside the whole tables."""
    dev = torch.device('cuda', 0)
    t = _tables(b, n0, n1, f, 5 + b)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    weights = {1: {k: (dv(t['w1'][k][0]), dv(t['w1'][k][1])) for k in TYPES}, 2: {k: (dv(t['w2'][k][0]), dv(t['w2'][k][1])) for k in TYPES}}
    g_emb = {k: dv(t['g_emb'][k]) for k in TYPES}

    def run(saved):
        grads = {}
        for layer, kdim in ((0, f), (1, 32), (2, 32)):
            for k in TYPES:
                grads[(layer, k)] = (torch.full((kdim, 16), 7.0, device=dev), torch.full((kdim, 16), 7.0, device=dev),
                                     torch.full((32,), 7.0, device=dev))
        kernels.Sage3BackwardAll(saved, weights, grads, n0, n1, True, 'linear').run(g_emb=g_emb)
        torch.cuda.synchronize()
        return grads

    whole = {name: {k: dv(t[name][k]) for k in TYPES} for name in ('z', 'h0', 'h1', 'x_mean0', 'x_mean1', 'x_mean2', 'row_ids')}
    whole['feat'] = {k: (dv(t['table'][k]), t['first']) for k in TYPES}
    want = run(whole)

    h1_end = b * (1 + n0)
    apart = {name: {k: dv(t[name][k][:b]) for k in TYPES} for name in ('z', 'h0', 'h1', 'x_mean0', 'x_mean1', 'x_mean2', 'row_ids')}
    apart['feat'] = whole['feat']
    apart['hop1'] = {name: {k: dv(t[src][k][b:h1_end]) for k in TYPES}
                     for name, src in (('h0', 'h0'), ('h1', 'h1'), ('x_mean0', 'x_mean0'), ('x_mean1', 'x_mean1'), ('ids', 'row_ids'))}
    apart['hop2'] = {'x_mean0': {k: dv(t['x_mean0'][k][h1_end:]) for k in TYPES}, 'ids': {k: dv(t['row_ids'][k][h1_end:]) for k in TYPES}}
    if with_mask:
        bits = {k: ((t['h0'][k][h1_end:] > 0).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(1, dtype=np.uint64).astype(np.uint32)
                for k in TYPES}
        apart['hop2']['mask'] = {k: dv(bits[k].view(np.int32)) for k in TYPES}
    else:
        apart['hop2']['h0'] = {k: dv(t['h0'][k][h1_end:]) for k in TYPES}
    got = run(apart)
    for key in want:
        for x, y in zip(got[key], want[key]):
            assert torch.equal(x, y), key
