"""Parity of the HIP path (through the C ABI) with the CPU oracle -- needs a GPU.

Integer work (sampling, regroup, gather) is compared bit for bit; floating-point
work (neighbor mean, projection, embeddings, gradients) within 1e-5 relative to
the float64 oracle, the tolerance the project states for fp32.
"""
import os

import numpy as np
import pytest
import torch

from conftest import golden_graph_parts
import graphsage_amd as ga
from graphsage_amd import kernels
from oracle import generator as ogen
from oracle import model as omodel
from oracle import walker

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def close(got, want, rtol=RTOL, scale=None):
    """Two bars, both must hold.  Per tensor: |got - want| <= rtol * max(|want|).  Per element:
    |got - want| <= rtol * |want| + 0.1 * rtol * max(|want|) -- 1e-5 relative for every element, with an
    absolute floor of 1e-6 of the tensor's largest magnitude, because an fp32 sum that cancels to
    (nearly) zero keeps the rounding error of its terms."""
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, dtype=np.float64)
    want = want.detach().cpu().double().numpy() if isinstance(want, torch.Tensor) else np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    s = scale if scale is not None else max(float(np.abs(want).max()), 1e-30)
    diff = np.abs(got - want)
    err = float(diff.max()) if diff.size else 0.0
    assert err <= rtol * s, "max abs err %.3e > %.1e * %.3e" % (err, rtol, s)
    tol = rtol * np.abs(want) + 0.1 * rtol * s
    bad = diff > tol
    assert not bad.any(), "%d of %d elements beyond %.0e relative (+ %.0e of the scale); worst err/tol %.2f" % (
        int(bad.sum()), bad.size, rtol, 0.1 * rtol, float((diff / tol).max()))


def csr_of(g):
    return np.asarray(g.indptr), np.asarray(g.indices)


# ---------------------------------------------------------- adjacency build ----

@pytest.mark.parametrize('name', ['g_small', 'g_mid'])
@pytest.mark.parametrize('chunk', [0, 1, 7, 1000])
def test_device_adjacency_build_equals_reference_neighbor_lists(golden, cuda, name, chunk):
    """gs_csr_build against the reference's neighbor_dict (flattened in the
    fixtures); chunk sizes below the edge count exercise the multi-pass append."""
    arrays, _ = golden
    feats, nodes, edges = golden_graph_parts(arrays, name)
    n = feats['user'].shape[0] + feats['item'].shape[0]
    a = torch.from_numpy(edges[0].astype(np.int32)).to(cuda)
    b = torch.from_numpy(edges[1].astype(np.int32)).to(cuda)
    indptr, indices = kernels.build_csr(a, b, n, symmetric=False, sort_chunk=chunk)
    np.testing.assert_array_equal(indptr.cpu().numpy(), arrays[name + '_adj_indptr'])
    np.testing.assert_array_equal(indices.cpu().numpy(), arrays[name + '_adj_indices'])
    # the fixture's edge array is [pairs, flipped pairs]: the symmetric form takes the first half
    half = edges.shape[1] // 2
    np.testing.assert_array_equal(edges[:, half:], edges[::-1, :half])
    indptr2, indices2 = kernels.build_csr(a[:half].contiguous(), b[:half].contiguous(), n, symmetric=True,
                                          sort_chunk=chunk)
    assert torch.equal(indptr2, indptr) and torch.equal(indices2, indices)
    g = ga.Graph.from_edges_device({k: torch.from_numpy(v).to(cuda) for k, v in feats.items()},
                                   a[:half].contiguous(), b[:half].contiguous())
    assert torch.equal(g.indptr, indptr) and g.num_users == feats['user'].shape[0]


def test_device_adjacency_build_edge_cases(cuda):
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=cuda)
    indptr, indices = kernels.build_csr(i32([]), i32([]), 5)
    assert indptr.tolist() == [0] * 6 and indices.numel() == 0
    indptr, indices = kernels.build_csr(i32([]), i32([]), 0)
    assert indptr.tolist() == [0]
    # duplicates kept, isolated nodes get empty rows, order = entry order
    indptr, indices = kernels.build_csr(i32([3, 0, 3, 3, 0]), i32([1, 4, 1, 0, 2]), 6)
    assert indptr.tolist() == [0, 2, 2, 2, 5, 5, 5] and indices.tolist() == [4, 2, 1, 1, 0]
    for bad in ([0, 6], [0, -1]):
        with pytest.raises(ValueError):
            kernels.build_csr(i32([0, 1]), i32(bad), 6)
        with pytest.raises(ValueError):
            kernels.build_csr(i32(bad), i32([0, 1]), 6, symmetric=True)
    # a large random case against numpy's stable sort, several chunk sizes, one hub row
    rng = np.random.default_rng(4)
    n, e = 5000, 300_000
    src = rng.integers(0, n, e).astype(np.int32)
    src[rng.integers(0, e, 60_000)] = 17
    dst = rng.integers(0, n, e).astype(np.int32)
    order = np.argsort(src, kind='stable')
    want_ptr = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n))])
    for chunk in (0, 4096, 99_999):
        indptr, indices = kernels.build_csr(torch.from_numpy(src).to(cuda), torch.from_numpy(dst).to(cuda), n,
                                            sort_chunk=chunk)
        np.testing.assert_array_equal(indptr.cpu().numpy(), want_ptr)
        np.testing.assert_array_equal(indices.cpu().numpy(), dst[order])


# ---------------------------------------------------------------- sampler ----

@pytest.mark.parametrize('case', range(7))
def test_sampler_replay_is_bit_exact_with_reference(golden, cuda, case):
    arrays, meta = golden
    info = meta['replay'][case]
    key, ns = info['key'], info['num_samples']
    g = ga.Graph(*golden_graph_parts(arrays, info['graph']))
    links = arrays[key + '_links']
    roots = np.concatenate([links[:, 0], links[:, 1]]).astype(np.int32)
    dev = g.device_tensors()
    tape = torch.from_numpy(arrays[key + '_tape'].view(np.int32)).to(cuda)
    hops, err = kernels.sample(dev['indptr'], dev['indices'], torch.from_numpy(roots).to(cuda), ns, replay=tape)
    assert err.tolist() == [0, 0]
    for h in range(len(ns)):
        np.testing.assert_array_equal(hops[h].cpu().numpy(), arrays[key + '_hop%d' % (h + 1)])


@pytest.mark.parametrize('case', range(3))
def test_walker_replay_returns_reference_walk_lists(golden, cuda, case):
    arrays, meta = golden
    info = meta['walker_only'][case]
    key, ns = info['key'], info['num_samples']
    g = ga.Graph(*golden_graph_parts(arrays, info['graph']))
    roots = [int(r) for r in arrays[key + '_roots']]
    w = ga.BreadthFirstWalker(g, ns)
    w.set_replay(arrays[key + '_tape'])
    walks = w.run_breadth_first_walk(roots)
    hops = walker.walks_to_hops(walks, ns)
    for h in range(len(ns)):
        np.testing.assert_array_equal(hops[h], arrays[key + '_hop%d' % (h + 1)])
    if case == 0:
        assert walks == meta['walk0_nested']


def test_walker_known_answer_walk(golden, cuda):
    """The reference's own walk for random.seed(7), replayed from the uniforms
    that seed produced (word = floor(u * 2**32); exact for these tiny degrees)."""
    arrays, meta = golden
    feats = {'user': arrays['kat_feat_user'], 'item': arrays['kat_feat_item']}
    g = ga.Graph(feats, {'user': [0, 1, 2], 'item': [3, 4]}, arrays['kat_edges'])
    w = ga.BreadthFirstWalker(g, [3, 2])
    w.set_replay(np.floor(np.asarray(meta['kat_uniforms_seed7']) * 2.0 ** 32).astype(np.uint32))
    assert w.run_breadth_first_walk([0, 4]) == meta['kat_walk_seed7']


@pytest.mark.parametrize('ns', [[25, 10], [4, 3, 2], [70, 3], [5], [3, 2, 2, 2]])
def test_sampler_philox_matches_oracle(golden, cuda, ns):
    arrays, _ = golden
    g = ga.Graph(*golden_graph_parts(arrays, 'g_mid'))
    indptr, indices = csr_of(g)
    rng = np.random.default_rng(5)
    roots = rng.integers(0, g.num_nodes, 301).astype(np.int32)
    dev = g.device_tensors()
    seed, base = 0xDEADBEEFCAFEF00D, (1 << 33) + 12345
    hops, err = kernels.sample(dev['indptr'], dev['indices'], torch.from_numpy(roots).to(cuda), ns,
                               rng_seed=seed, walk_id_base=base)
    assert err.tolist() == [0, 0]
    want = walker.walk_vectorised(indptr, indices, roots, ns, rng_seed=seed, first_walk=base)
    for got, w in zip(hops, want):
        np.testing.assert_array_equal(got.cpu().numpy(), w)
    # explicit walk ids: draws follow the id, not the position in the batch
    perm = rng.permutation(len(roots))
    ids = torch.from_numpy((base + perm).astype(np.int64)).to(cuda)
    hops2, _ = kernels.sample(dev['indptr'], dev['indices'], torch.from_numpy(roots[perm]).to(cuda), ns,
                              rng_seed=seed, walk_ids=ids)
    for got, w in zip(hops2, want):
        np.testing.assert_array_equal(got.cpu().numpy(), w[perm])


def test_sampler_isolated_node_and_bad_seed(golden, cuda):
    edges = np.array([[0, 1, 3, 3], [3, 3, 0, 1]])
    feats = {'user': np.zeros((3, 2), np.float32), 'item': np.zeros((2, 2), np.float32)}
    g = ga.Graph(feats, {'user': [0, 1, 2], 'item': [3, 4]}, edges)
    w = ga.BreadthFirstWalker(g, [2, 2])
    with pytest.raises(KeyError):
        w.run_breadth_first_walk([2])
    with pytest.raises(ValueError) as ei:
        w.run_breadth_first_walk([0, 2])
    assert str(ei.value).startswith("Samples must exist. Self-node also can be a sample")
    assert w.run_breadth_first_walk([]) == []
    with pytest.raises(KeyError):
        w.run_breadth_first_walk([7])
    dev = g.device_tensors()
    hops, err = kernels.sample(dev['indptr'], dev['indices'],
                               torch.tensor([0, 2, 99, -4], dtype=torch.int32, device=cuda), [2, 2])
    assert err.tolist() == [3, 1]                      # 1 + isolated node 2; an out-of-range seed was seen
    assert (hops[0][1:] == -1).all() and (hops[1][1:] == -1).all() and (hops[0][0] >= 0).all()
    # empty input
    hops, err = kernels.sample(dev['indptr'], dev['indices'], torch.zeros(0, dtype=torch.int32, device=cuda), [2, 2])
    assert hops[0].shape == (0, 2) and hops[1].shape == (0, 4)


@pytest.mark.parametrize('case', range(7))
def test_generator_batches_equal_reference_arrays(golden, cuda, case):
    """flow()/sample_features on replayed draws: the six slot tensors equal the
    arrays the reference generator produced, element for element."""
    arrays, meta = golden
    info = meta['replay'][case]
    key, ns, b = info['key'], info['num_samples'], info['batch']
    g = ga.Graph(*golden_graph_parts(arrays, info['graph']))
    links = [tuple(int(v) for v in l) for l in arrays[key + '_links']]
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'])
    gen.sampler.set_replay(arrays[key + '_tape'])
    batch = gen.sample_features(links, 0)
    assert len(batch) == 2 * (len(ns) + 1)
    for i in range(len(batch)):
        want = arrays[key + '_x%d' % i]
        assert tuple(batch[i].shape) == want.shape
        np.testing.assert_array_equal(batch[i].cpu().numpy(), want)
    # through flow(): same thing, targets ride along
    gen.sampler.set_replay(arrays[key + '_tape'])
    seq = gen.flow(links, targets=np.arange(b), shuffle=False)
    feats, targets = seq[0]
    np.testing.assert_array_equal(targets, np.arange(b))
    np.testing.assert_array_equal(feats[len(feats) - 1].cpu().numpy(), arrays[key + '_x%d' % (len(feats) - 1)])


def test_generator_known_answer_shapes_and_values(golden, cuda):
    arrays, meta = golden
    feats = {'user': arrays['kat_feat_user'], 'item': arrays['kat_feat_item']}
    g = ga.Graph(feats, {'user': [0, 1, 2], 'item': [3, 4]}, arrays['kat_edges'])
    gen = ga.GraphSAGEGenerator(g, 2, [3, 2], head_node_types=['user', 'item'])
    import random
    random.seed(3)
    us = [random.random() for _ in range(2 * 2 * 9)]
    gen.sampler.set_replay(np.floor(np.asarray(us) * 2.0 ** 32).astype(np.uint32))
    seq = gen.flow([(0, 3), (2, 4), (1, 3)], targets=[1, 0, 1])
    assert len(seq) == meta['kat_gen_seed3_len']
    f0, t0 = seq[0]
    assert [list(f0[i].shape) for i in range(6)] == meta['kat_gen_seed3_shapes']
    assert [int(t) for t in t0] == meta['kat_gen_seed3_targets']
    for i in range(6):
        np.testing.assert_array_equal(f0[i].cpu().numpy(), arrays['kat_gen_seed3_x%d' % i].astype(np.float32))


def test_sampling_is_independent_of_batch_split_and_rank(golden, cuda):
    """Draws are keyed by (link position, side, hop, slot): the same link gets
    the same neighborhood whether it is sampled in one batch, in small batches
    or on another rank of a data-parallel run."""
    arrays, _ = golden
    g = ga.Graph(*golden_graph_parts(arrays, 'g_mid'))
    e = arrays['g_mid_edges']
    links = [(int(e[0, i]), int(e[1, i])) for i in range(40)]
    targets = np.arange(40)

    def collect(batch_size, rank=0, world=1):
        gen = ga.GraphSAGEGenerator(g, batch_size, [5, 3], head_node_types=['user', 'item'], seed=11)
        seq = gen.flow(links, targets=targets, shuffle=True, seed=3, rank=rank, world_size=world)
        out = {}
        for feats, t in seq:
            ids = [x.cpu().numpy() for x in feats.ids]
            for r, link in enumerate(t):
                out[int(link)] = [a[r].copy() for a in ids]
        return out

    whole = collect(40)
    small = collect(7)
    r0, r1 = collect(8, 0, 2), collect(8, 1, 2)
    assert sorted(list(r0) + list(r1)) == list(range(40)) and not set(r0) & set(r1)
    for link in range(40):
        other = r0.get(link) or r1.get(link)
        for a, b, c in zip(whole[link], small[link], other):
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(a, c)
    # a new epoch resamples
    gen = ga.GraphSAGEGenerator(g, 40, [5, 3], seed=11)
    seq = gen.flow(links, targets=targets, shuffle=False)
    a = seq[0][0].ids[4].cpu().numpy()
    seq.on_epoch_end()
    assert not np.array_equal(a, seq[0][0].ids[4].cpu().numpy())


# ------------------------------------------------------------ gather-mean ----

@pytest.mark.parametrize('f', [64, 24, 19, 30, 18, 256, 300, 1])
@pytest.mark.parametrize('n_child', [10, 25, 5, 1, 9])
def test_gather_rows_and_gather_mean(cuda, f, n_child):
    rng = np.random.default_rng(f * 100 + n_child)
    n_rows, n_parents, off = 777, 133, 1000
    feat = rng.standard_normal((n_rows, f)).astype(np.float32)
    ids = rng.integers(0, n_rows, n_parents * n_child).astype(np.int32) + off
    ids[3] = -1                                           # below an isolated node: contributes zeros
    ids[5] = off + n_rows                                 # out of range on the other side
    d_feat, d_ids = torch.from_numpy(feat).to(cuda), torch.from_numpy(ids).to(cuda)
    rows = kernels.gather_rows(d_feat, d_ids, off).cpu().numpy()
    valid = (ids - off >= 0) & (ids - off < n_rows)
    want_rows = np.where(valid[:, None], feat[np.clip(ids - off, 0, n_rows - 1)], 0).astype(np.float32)
    np.testing.assert_array_equal(rows, want_rows)        # a copy: bit exact
    got = kernels.gather_mean(d_feat, d_ids, n_parents, n_child, off)
    want = want_rows.astype(np.float64).reshape(n_parents, n_child, f).mean(axis=1)
    close(got, want, scale=1.0)
    # dense groups give the same bits as the indexed form on materialised rows
    dense = kernels.gather_mean(torch.from_numpy(want_rows).to(cuda), None, n_parents, n_child)
    assert torch.equal(dense, got)
    # empty input
    assert kernels.gather_mean(d_feat, d_ids[:0], 0, n_child, off).shape == (0, f)
    assert kernels.gather_rows(d_feat, d_ids[:0], off).shape == (0, f)


@pytest.mark.parametrize('f,n_child', [(64, 10), (19, 25), (30, 9), (256, 1), (1, 70)])
def test_gather_max_indexed_and_dense(cuda, f, n_child):
    rng = np.random.default_rng(f * 31 + n_child)
    n_rows, n_parents, off = 500, 77, 1000
    feat = rng.standard_normal((n_rows, f)).astype(np.float32)
    ids = rng.integers(0, n_rows, (n_parents, n_child)).astype(np.int32)
    t = torch.from_numpy(feat).to(cuda)
    got = kernels.gather_max(t, torch.from_numpy(ids + off).to(cuda).reshape(-1), n_parents, n_child, off)
    np.testing.assert_array_equal(got.cpu().numpy(), feat[ids].max(axis=1))
    dense = torch.from_numpy(feat[:(n_rows // n_child) * n_child]).to(cuda)
    got = kernels.gather_max(dense, None, n_rows // n_child, n_child)
    np.testing.assert_array_equal(got.cpu().numpy(), feat[:(n_rows // n_child) * n_child].reshape(-1, n_child, f).max(axis=1))
    # all-negative rows: the identity of the reduction must not leak in (zeros would win)
    neg = -np.abs(feat) - 1.0
    got = kernels.gather_max(torch.from_numpy(neg).to(cuda), torch.from_numpy(ids).to(cuda).reshape(-1), n_parents, n_child)
    np.testing.assert_array_equal(got.cpu().numpy(), neg[ids].max(axis=1))


def test_gather_mean_many_children_and_unaligned_views(cuda):
    rng = np.random.default_rng(1)
    feat = rng.standard_normal((50, 64)).astype(np.float32)
    ids = rng.integers(0, 50, 7 * 1000).astype(np.int32)           # child tile too large for LDS staging
    d = torch.from_numpy(feat).to(cuda)
    got = kernels.gather_mean(d, torch.from_numpy(ids).to(cuda), 7, 1000)
    close(got, feat[ids].astype(np.float64).reshape(7, 1000, 64).mean(1), scale=1.0)
    # a table whose base is only 4-byte aligned falls back to scalar lanes
    buf = torch.zeros(50 * 64 + 1, dtype=torch.float32, device=cuda)
    buf[1:] = d.reshape(-1)
    view = buf[1:].reshape(50, 64)
    got2 = kernels.gather_mean(view, torch.from_numpy(ids).to(cuda), 7, 1000)
    close(got2, got, rtol=1e-6, scale=1.0)


def test_gather_mean_backward_scatter_add(cuda):
    rng = np.random.default_rng(2)
    n_rows, f, n_parents, n_child, off = 40, 24, 64, 10, 7
    ids = rng.integers(0, n_rows, n_parents * n_child).astype(np.int32) + off
    g = rng.standard_normal((n_parents, f)).astype(np.float32)
    want = np.zeros((n_rows, f))
    np.add.at(want, ids - off, np.repeat(g.astype(np.float64) / n_child, n_child, axis=0))
    g_feat = torch.zeros((n_rows, f), dtype=torch.float32, device=cuda)
    kernels.gather_mean_bwd(torch.from_numpy(g).to(cuda), torch.from_numpy(ids).to(cuda), n_child, g_feat, off)
    close(g_feat, want, scale=float(np.abs(want).max()))
    dense = torch.empty((n_parents * n_child, f), dtype=torch.float32, device=cuda)
    kernels.gather_mean_bwd(torch.from_numpy(g).to(cuda), None, n_child, dense)
    close(dense, np.repeat(g.astype(np.float64) / n_child, n_child, axis=0))


# -------------------------------------------------------------- projection ----

SHAPES = [(213, 64, 64, 32), (200, 24, 19, 32), (200, 19, 24, 32), (1000, 32, 32, 32),
          (77, 256, 256, 256), (65, 30, 18, 6), (1, 64, 64, 32), (130, 7, 3, 2), (64, 64, 64, 160),
          (700, 256, 256, 256), (1333, 64, 128, 256), (512, 32, 256, 512),      # these three take the 128x128-tile kernel
          (5000, 128, 256, 256), (4100, 256, 128, 512), (31, 128, 128, 256)]    # and these its weight-gradient twin


@pytest.mark.parametrize('m,fs,fn,h', SHAPES)
@pytest.mark.parametrize('act', ['relu', 'linear'])
def test_projection_forward_and_backward(cuda, m, fs, fn, h, act):
    gen = torch.Generator().manual_seed(m + fs + h)
    xs = torch.randn(m, fs, generator=gen, dtype=torch.float64)
    xm = torch.randn(m, fn, generator=gen, dtype=torch.float64)
    ws = omodel.glorot_uniform(fs, h // 2, gen, torch.float64).requires_grad_(True)
    wn = omodel.glorot_uniform(fn, h // 2, gen, torch.float64).requires_grad_(True)
    bias = (torch.randn(h, generator=gen, dtype=torch.float64) * 0.1).requires_grad_(True)
    xs.requires_grad_(True)
    xm.requires_grad_(True)
    want = omodel.aggregator(xs[:, None, :], xm[:, None, None, :], ws, wn, bias, act)[:, 0, :]
    g = torch.randn(m, h, generator=gen, dtype=torch.float64)
    if act == 'relu':
        # an output whose pre-activation lies within fp32 rounding of zero may be masked either way:
        # such elements get no upstream gradient, everything else is compared
        with torch.no_grad():
            pre = torch.cat([xs @ ws, xm @ wn], dim=1) + bias
        g = g * (pre.abs() > 1e-6)
    want.backward(g)

    f32 = lambda t: t.detach().float().to(cuda).contiguous()
    out = kernels.sage_project(f32(xs), f32(xm), f32(ws), f32(wn), f32(bias), act)
    scale = float(want.abs().max())
    close(out, want, scale=scale)
    g_ws, g_wn, g_b, g_xs, g_xm = kernels.sage_project_bwd(f32(xs), f32(xm), f32(ws), f32(wn), act, out, f32(g))
    close(g_ws, ws.grad)
    close(g_wn, wn.grad)
    close(g_b, bias.grad)
    close(g_xs, xs.grad, scale=float(xs.grad.abs().max()) + 1e-30)
    close(g_xm, xm.grad, scale=float(xm.grad.abs().max()) + 1e-30)
    # no bias, skipped input gradients
    out_nb = kernels.sage_project(f32(xs), f32(xm), f32(ws), f32(wn), None, act)
    close(out_nb, omodel.aggregator(xs[:, None, :], xm[:, None, None, :], ws, wn, None, act)[:, 0, :], scale=scale)
    r = kernels.sage_project_bwd(f32(xs), f32(xm), f32(ws), f32(wn), act, out, f32(g), need_x_self=False,
                                 need_x_mean=False, need_bias=False)
    assert r[2] is None and r[3] is None and r[4] is None
    assert torch.equal(r[0], g_ws) and torch.equal(r[1], g_wn)     # slab reduction: reproducible bits


def test_projection_rejects_bad_shapes(cuda):
    z = lambda *s: torch.zeros(*s, device=cuda)
    with pytest.raises(ValueError):
        kernels.sage_project(z(4, 8), z(5, 8), z(8, 4), z(8, 4), None, 'relu')
    with pytest.raises(ValueError):
        kernels.sage_project(z(4, 8), z(4, 8), z(7, 4), z(8, 4), None, 'relu')
    with pytest.raises(TypeError):
        kernels.sage_project(z(4, 8).double(), z(4, 8), z(8, 4), z(8, 4), None, 'relu')
    with pytest.raises(Exception):
        kernels.sage_project(z(4, 8).cpu(), z(4, 8), z(8, 4), z(8, 4), None, 'relu')
    assert kernels.sage_project(z(0, 8), z(0, 8), z(8, 4), z(8, 4), None, 'relu').shape == (0, 8)


def test_l2_normalize_forward_backward(cuda):
    gen = torch.Generator().manual_seed(0)
    for m, h in [(100, 32), (3, 256), (5, 1), (64, 70)]:
        x = torch.randn(m, h, generator=gen, dtype=torch.float64)
        x[0] = 0                                      # the eps branch
        x.requires_grad_(True)
        want = omodel.l2_normalize(x)
        g = torch.randn(m, h, generator=gen, dtype=torch.float64)
        want.backward(g)
        xf = x.detach().float().to(cuda)
        close(kernels.l2_normalize(xf), want, scale=1.0)
        close(kernels.l2_normalize_bwd(xf, g.float().to(cuda)), x.grad, scale=float(x.grad.abs().max()))


# ------------------------------------------------------------------- model ----

def _oracle_model(model, plan):
    weights = {}
    for layer, aggs in enumerate(model._aggs):
        for node_type, agg in aggs.items():
            weights[(layer, node_type)] = {
                'w_self': agg.w_self.detach().cpu().double().requires_grad_(True),
                'w_neigh': agg.w_neigh.detach().cpu().double().requires_grad_(True),
                'bias': None if agg.bias is None else agg.bias.detach().cpu().double().requires_grad_(True)}
    return weights


@pytest.mark.parametrize('ns,sizes,gname,b', [([8, 4], [32, 32], 'g_mid', 50), ([25, 10], [32, 32], 'g_mid', 9),
                                              ([4, 3, 2], [16, 8, 6], 'g_small', 7), ([3], [4], 'g_small', 5)])
@pytest.mark.parametrize('agg', ['mean', 'max'])
def test_model_embeddings_and_gradients_match_oracle(golden, cuda, ns, sizes, gname, b, agg):
    arrays, _ = golden
    g = ga.Graph(*golden_graph_parts(arrays, gname))
    e = arrays[gname + '_edges']
    links = [(int(e[0, i]), int(e[1, i])) for i in range(b)]
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=4)
    model = ga.GraphSAGE(sizes, gen, agg_option=agg, seed=9).build()
    for p in model.parameters():                      # biases away from zero so they matter
        if p.dim() == 1:
            p.data.uniform_(-0.1, 0.1)
    batch = gen.sample_features(links, 0)

    plan = omodel.LayerPlan(sizes, ns, g.node_feature_sizes)
    weights = _oracle_model(model, plan)
    xin = [x.double() for x in map(torch.from_numpy, batch.numpy())]
    want = omodel.forward(plan, weights, xin, agg=agg)

    fused = model(batch)                              # ids -> gather-mean, deepest slots never materialised
    dense = model([batch[i] for i in range(len(batch))])
    assert len(fused) == 2 and tuple(fused[0].shape) == (b, sizes[-1])
    for k in range(2):
        close(fused[k], want[k], scale=1.0)
        assert torch.equal(fused[k], dense[k])         # both paths sum in the same order

    # gradients of a scalar loss through every kernel's backward
    gen_t = torch.Generator().manual_seed(1)
    cu, ci = torch.randn(b, sizes[-1], generator=gen_t, dtype=torch.float64), torch.randn(b, sizes[-1], generator=gen_t, dtype=torch.float64)
    ((want[0] * cu).sum() + (want[1] * ci).sum()).backward()
    ((fused[0] * cu.float().to(cuda)).sum() + (fused[1] * ci.float().to(cuda)).sum()).backward()
    for (layer, node_type), w in weights.items():
        agg = model._aggs[layer][node_type]
        for name, p in (('w_self', agg.w_self), ('w_neigh', agg.w_neigh), ('bias', agg.bias)):
            ref = w[name].grad
            close(p.grad, ref, rtol=2e-5, scale=float(ref.abs().max()) + 1e-30)


def test_aggregator_layer_call_convention(cuda):
    gen = torch.Generator().manual_seed(3)
    x_self = torch.randn(6, 4, 24, generator=gen)
    x_neigh = torch.randn(6, 4, 8, 19, generator=gen)
    agg = ga.Aggregator(32, seed=5)
    out = agg([x_self.to(cuda), x_neigh.to(cuda)])
    assert tuple(out.shape) == (6, 4, 32)
    assert tuple(agg.w_self.shape) == (24, 16) and tuple(agg.w_neigh.shape) == (19, 16) and tuple(agg.bias.shape) == (32,)
    assert float(agg.bias.abs().max()) == 0.0
    limit = (6.0 / (24 + 16)) ** 0.5
    assert float(agg.w_self.abs().max()) <= limit
    want = omodel.aggregator(x_self.double(), x_neigh.double(), agg.w_self.detach().cpu().double(),
                             agg.w_neigh.detach().cpu().double(), agg.bias.detach().cpu().double(), 'relu')
    close(out, want)
    # the other aggregation of the reference
    agg = ga.Aggregator(32, agg_option='max', seed=5)
    x_neigh[0, 0, 3] = x_neigh[0, 0, 5]                   # tied maxima share the gradient
    xs, xn = x_self.to(cuda).requires_grad_(True), x_neigh.to(cuda).requires_grad_(True)
    out = agg([xs, xn])
    os_, on_ = x_self.double().requires_grad_(True), x_neigh.double().requires_grad_(True)
    want = omodel.aggregator(os_, on_, agg.w_self.detach().cpu().double(), agg.w_neigh.detach().cpu().double(),
                             agg.bias.detach().cpu().double(), 'relu', agg='max')
    close(out, want)
    cot = torch.randn(6, 4, 32, generator=gen)
    (out * cot.to(cuda)).sum().backward()
    (want * cot.double()).sum().backward()
    close(xn.grad, on_.grad, scale=float(on_.grad.abs().max()))
    close(xs.grad, os_.grad, scale=float(os_.grad.abs().max()))


# ------------------------------------------------- fused layer and engine ----

@pytest.mark.parametrize('fs,fn,h', [(64, 64, 32), (32, 32, 32), (64, 64, 64), (128, 64, 32), (32, 128, 64),
                                     (256, 256, 256), (64, 128, 256), (128, 96, 512)])   # the last three: wide kernel
@pytest.mark.parametrize('act,normalize', [('relu', False), ('linear', True)])
@pytest.mark.parametrize('pieces', [True, False])
def test_fused_layer_matches_oracle(cuda, fs, fn, h, act, normalize, pieces):
    """``pieces``: wide layers have two forms of the matrix product (bf16 pieces of the fp32 operands on the
    matrix cores / the fp32 matrix instruction); narrow layers have one and ignore the switch."""
    if not pieces and h < 256:
        pytest.skip("one form only")
    rng = np.random.default_rng(fs + fn + h)
    gen = torch.Generator().manual_seed(fs * 7 + h)
    n_self, n_neigh, off_s, off_n = 500, 300, 11, 1000
    t_self = rng.standard_normal((n_self, fs)).astype(np.float32)
    t_neigh = rng.standard_normal((n_neigh, fn)).astype(np.float32)
    segs_np = []
    for n_heads, n_child in [(37, 10), (1, 25), (130, 3)]:
        heads = rng.integers(0, n_self, n_heads).astype(np.int32) + off_s
        kids = rng.integers(0, n_neigh, n_heads * n_child).astype(np.int32) + off_n
        segs_np.append((heads, kids, n_child))
    segs_np[0][1][4] = -1                                     # sample below an isolated node
    segs_np[2][0][7] = off_s + n_self                         # head outside the table
    ws = omodel.glorot_uniform(fs, h // 2, gen, torch.float64)
    wn = omodel.glorot_uniform(fn, h // 2, gen, torch.float64)
    bias = torch.randn(h, generator=gen, dtype=torch.float64) * 0.1

    def rows_of(table, ids, off):
        ok = (ids - off >= 0) & (ids - off < len(table))
        return np.where(ok[:, None], table[np.clip(ids - off, 0, len(table) - 1)], 0).astype(np.float64)

    want, want_self, want_mean = [], [], []
    for heads, kids, n_child in segs_np:
        xs = torch.from_numpy(rows_of(t_self, heads, off_s))
        xn = torch.from_numpy(rows_of(t_neigh, kids, off_n)).reshape(len(heads), 1, n_child, fn)
        o = omodel.aggregator(xs[:, None, :], xn, ws, wn, bias, act)[:, 0, :]
        want.append(omodel.l2_normalize(o) if normalize else o)
        want_self.append(xs)
        want_mean.append(xn.mean(2)[:, 0, :])
    want = torch.cat(want)

    d = lambda a: torch.from_numpy(a).to(cuda)
    f32 = lambda t: t.float().to(cuda).contiguous()
    segs = [{'head_ids': d(hh), 'child_ids': d(kk), 'n_heads': len(hh), 'n_child': nc} for hh, kk, nc in segs_np]
    rows = sum(len(s[0]) for s in segs_np)
    save_s = torch.empty((rows, fs), device=cuda)
    save_m = torch.empty((rows, fn), device=cuda)
    assert kernels.fused_supported(fs, fn, h, 25)
    got = kernels.sage_fused(d(t_self), off_s, d(t_neigh), off_n, segs, f32(ws), f32(wn), f32(bias), act,
                             normalize=normalize, save_self=save_s, save_mean=save_m, exact_pieces=pieces)
    close(got, want, scale=float(want.abs().max()))
    np.testing.assert_array_equal(save_s.cpu().numpy(), torch.cat(want_self).float().numpy())
    close(save_m, torch.cat(want_mean), scale=1.0)

    # dense addressing (layer >= 1): heads and children are rows of activation buffers
    n_heads, n_child = 45, 5
    dense = [{'n_heads': n_heads, 'n_child': n_child, 'self_row0': 3, 'child_row0': 10}]
    got2 = kernels.sage_fused(d(t_self), 0, d(t_neigh), 0, dense, f32(ws), f32(wn), None, act, normalize=normalize,
                              exact_pieces=pieces)
    xs = torch.from_numpy(t_self[3:3 + n_heads].astype(np.float64))
    xn = torch.from_numpy(t_neigh[10:10 + n_heads * n_child].astype(np.float64)).reshape(n_heads, 1, n_child, fn)
    o = omodel.aggregator(xs[:, None, :], xn, ws, wn, None, act)[:, 0, :]
    close(got2, omodel.l2_normalize(o) if normalize else o, scale=float(o.abs().max()))
    with pytest.raises(Exception):                            # dense rows beyond the table
        kernels.sage_fused(d(t_self), 0, d(t_neigh), 0, [{'n_heads': 100, 'n_child': 5, 'child_row0': 0}],
                           f32(ws), f32(wn), None, act)
    assert not kernels.fused_supported(24, 19, 32, 8) and not kernels.fused_supported(256, 256, 256, 40)
    assert not kernels.fused_supported(48, 64, 256, 5) and kernels.fused_supported(256, 256, 256, 5)


@pytest.mark.parametrize('pieces', [True, False])
@pytest.mark.parametrize('fs,fn,h,n_child', [(64, 96, 256, 5), (256, 256, 256, 12), (64, 64, 256, 32)])
def test_wide_fused_layer_many_tiles_equals_separate_kernels(cuda, fs, fn, h, n_child, pieces):
    """More 128-head tiles than blocks: every block walks several tiles (ids staged one
    tile ahead, operand stages alternating across tile borders).  Checked against gather +
    gather-mean + projection, and a sample of rows against the float64 oracle."""
    rng = np.random.default_rng(fs + n_child)
    gen = torch.Generator().manual_seed(h + n_child)
    n_self, n_neigh, off_s, off_n = 5000, 3000, 7, 6000
    t_self = torch.from_numpy(rng.standard_normal((n_self, fs)).astype(np.float32)).to(cuda)
    t_neigh = torch.from_numpy(rng.standard_normal((n_neigh, fn)).astype(np.float32)).to(cuda)
    ws = omodel.glorot_uniform(fs, h // 2, gen, torch.float64)
    wn = omodel.glorot_uniform(fn, h // 2, gen, torch.float64)
    bias = torch.randn(h, generator=gen, dtype=torch.float64) * 0.1
    f32 = lambda t: t.float().to(cuda).contiguous()
    segs, want = [], []
    for n_heads, nc in [(300 * 128 + 5, n_child), (77, 3), (2 * 128, 1)]:
        heads = torch.from_numpy(rng.integers(0, n_self, n_heads).astype(np.int32) + off_s).to(cuda)
        kids = torch.from_numpy(rng.integers(0, n_neigh, n_heads * nc).astype(np.int32) + off_n).to(cuda)
        # samples below isolated nodes: in a third of the large segment's tiles (the others take the path
        # that checks no key), in all of the small segment, in none of the last
        if nc > 1:
            kids[:len(kids) // 3 if n_heads > 1000 else len(kids):97] = -1
        segs.append({'head_ids': heads, 'child_ids': kids, 'n_heads': n_heads, 'n_child': nc})
        xs = kernels.gather_rows(t_self, heads, off_s)
        xm = kernels.gather_mean(t_neigh, kids, n_heads, nc, off_n)
        want.append(kernels.sage_project(xs, xm, f32(ws), f32(wn), f32(bias), 'relu'))
    want = torch.cat(want)
    rows = want.shape[0]
    save_s, save_m = torch.empty((rows, fs), device=cuda), torch.empty((rows, fn), device=cuda)
    got = kernels.sage_fused(t_self, off_s, t_neigh, off_n, segs, f32(ws), f32(wn), f32(bias), 'relu',
                             save_self=save_s, save_mean=save_m, exact_pieces=pieces)
    close(got, want, scale=float(want.abs().max()))
    # rows of the first tiles, a tile border in the middle, and the tail against the float64 restatement
    pick = torch.cat([torch.arange(0, 200), torch.arange(128 * 257 - 20, 128 * 257 + 20), torch.arange(rows - 400, rows)])
    heads_all = torch.cat([s_['head_ids'] for s_ in segs]).cpu().numpy()
    ts, tn = t_self.cpu().numpy().astype(np.float64), t_neigh.cpu().numpy().astype(np.float64)
    start = 0
    for s_ in segs:
        nh, nc = s_['n_heads'], s_['n_child']
        sel = pick[(pick >= start) & (pick < start + nh)].numpy() - start
        if len(sel):
            hid = s_['head_ids'].cpu().numpy()[sel] - off_s
            kid = s_['child_ids'].cpu().numpy().reshape(nh, nc)[sel] - off_n
            ok = (kid >= 0) & (kid < n_neigh)
            xn = np.where(ok[..., None], tn[np.clip(kid, 0, n_neigh - 1)], 0.0)
            o = omodel.aggregator(torch.from_numpy(ts[hid])[:, None, :], torch.from_numpy(xn)[:, None], ws, wn, bias, 'relu')[:, 0, :]
            close(got[start + sel], o, scale=float(o.abs().max()))
            np.testing.assert_array_equal(save_s[start + sel].cpu().numpy(), ts[hid].astype(np.float32))
            close(save_m[start + sel], torch.from_numpy(xn.mean(1)), scale=1.0)
        start += nh


@pytest.mark.parametrize('pieces', [True, False])
@pytest.mark.parametrize('shape', [
    [(4096, 25), (102400 // 4, 10), (1024000 // 16, 5)],                          # the three-hop mix: few expensive tiles first
    [(40 * 128, 1), (3, 32), (700 * 128 + 1, 2), (128, 17), (129, 9), (5000, 4), (1, 1), (260 * 128, 3)],   # eight segments
    [(100, 7), (90, 3)],                                                         # fewer tiles than blocks
    [(256 * 128 * 3, 2)],                                                        # one segment, whole rounds
])
def test_wide_layer_every_tile_is_taken_once_whatever_the_mix_of_segments(cuda, shape, pieces):
    """The blocks of the wide kernel share the tiles by estimated work (expensive segments first, classes of
    blocks with equal shares): every head of every segment must come out, and equal gather + gather-mean +
    projection, for segment mixes that give one class, several classes, empty shares and fewer tiles than blocks."""
    fs = fn = 64
    h = 256
    rng = np.random.default_rng(len(shape) * 1000 + shape[0][0])
    gen = torch.Generator().manual_seed(len(shape))
    n_self, n_neigh = 4000, 3000
    t_self = torch.from_numpy(rng.standard_normal((n_self, fs)).astype(np.float32)).to(cuda)
    t_neigh = torch.from_numpy(rng.standard_normal((n_neigh, fn)).astype(np.float32)).to(cuda)
    f32 = lambda t: t.float().to(cuda).contiguous()
    ws, wn = f32(omodel.glorot_uniform(fs, h // 2, gen, torch.float64)), f32(omodel.glorot_uniform(fn, h // 2, gen, torch.float64))
    bias = f32(torch.randn(h, generator=gen, dtype=torch.float64) * 0.1)
    segs, want = [], []
    for n_heads, nc in shape:
        heads = torch.from_numpy(rng.integers(0, n_self, n_heads).astype(np.int32)).to(cuda)
        kids = torch.from_numpy(rng.integers(0, n_neigh, n_heads * nc).astype(np.int32)).to(cuda)
        segs.append({'head_ids': heads, 'child_ids': kids, 'n_heads': n_heads, 'n_child': nc})
        want.append(kernels.sage_project(kernels.gather_rows(t_self, heads, 0), kernels.gather_mean(t_neigh, kids, n_heads, nc, 0),
                                         ws, wn, bias, 'relu'))
    want = torch.cat(want)
    out = torch.full_like(want, float('nan'))                # a tile nobody takes stays NaN
    got = kernels.sage_fused(t_self, 0, t_neigh, 0, segs, ws, wn, bias, 'relu', out=out, exact_pieces=pieces)
    assert not bool(torch.isnan(got).any())
    close(got, want, scale=float(want.abs().max()))


def test_wide_kernel_wait_counts(cuda, tmp_path):
    """The wide kernel's producers wait for their row loads by count (``s_waitcnt vmcnt(N)`` with N the number
    of loads issued after the ones waited for).  A wrong count reads registers before their data arrives and
    faults nothing.  libgsage_safewait.so is the same source with every such wait replaced by a wait for all
    loads: both builds must produce the same bits over full and partial chunks, single children, zero rows,
    several segments and saved copies."""
    import subprocess
    import sys
    from conftest import ROOT
    safe = os.path.join(ROOT, 'graphsage_amd', 'libgsage_safewait.so')
    assert os.path.exists(safe), "build the checking library: make -C graphsage_amd/csrc safewait"
    worker = os.path.join(ROOT, 'tests', 'wide_waits_worker.py')
    res = {}
    for name, lib in (('product', None), ('safe', safe)):
        env = {k: v for k, v in os.environ.items() if k != 'GSAGE_LIB'}
        if lib:
            env['GSAGE_LIB'] = lib
        out = str(tmp_path / (name + '.pt'))
        r = subprocess.run([sys.executable, worker, '--out', out], env=env, capture_output=True, text=True, timeout=200, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert (lib or 'libgsage.so') in r.stdout
        res[name] = torch.load(out, weights_only=True)
    assert res['product'].keys() == res['safe'].keys() and len(res['product']) >= 10
    for k in res['product']:
        a, b = res['product'][k], res['safe'][k]
        assert not torch.isnan(a).any(), k
        assert torch.equal(a, b), "%s differs between the counted and the full waits" % k


@pytest.mark.parametrize('ns,sizes,fu,fi,b', [([6, 4, 3], [32, 32, 32], 64, 64, 24), ([25, 10, 5], [32, 32, 32], 64, 64, 40),
                                                ([3, 5, 2], [64, 64, 32], 64, 64, 130), ([4, 3, 2], [32, 32, 64], 32, 32, 7),
                                                ([5, 7, 4], [32, 32, 32], 128, 128, 33), ([9, 2, 6], [32, 32, 32], 64, 32, 21),
                                                ([2, 11, 1], [32, 32, 64], 32, 64, 50), ([30, 1, 12], [32, 32, 32], 64, 64, 3)])
def test_three_hop_engine_through_the_subtree_kernel(cuda, ns, sizes, fu, fi, b):
    """Three hops in two launches (gs_sage2_forward_subtrees drawing all three hops, gs_sage3_seeds_forward for
    the seeds): the ids are gs_sample's, with Philox keys and with a replay tape, and the embeddings equal
    those of the layer-by-layer kernels."""
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(b)
    nu, ni = 400, 90
    u = np.concatenate([rng.integers(0, nu, 3000), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 3000), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, fu)).astype(np.float32),
             'item': rng.standard_normal((ni, fi)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=4)
    model = ga.GraphSAGE(sizes, gen, seed=6).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.1, 0.1)
    seeds = torch.tensor(list(e[0, :b]) + list(e[1, :b]), dtype=torch.int32, device=cuda)
    walk_ids = torch.arange(100, 100 + 2 * b, dtype=torch.int64, device=cuda)
    eng_s = SageForwardEngine(g, model, b, keep_ids=True, subtree_kernel=True)
    eng_g = SageForwardEngine(g, model, b, subtree_kernel=False)       # sampler + one fused kernel per layer and type
    assert eng_s.subtree_kernel and not eng_g.subtree_kernel
    tape = torch.from_numpy(rng.integers(0, 2 ** 32, 2 * b * sum(eng_s.sizes[1:]), dtype=np.uint64).astype(np.uint32)
                            .view(np.int32)).to(cuda)
    for kw in ({'rng_seed': 77}, {'rng_seed': 0, 'replay': tape}):
        out_s = [x.clone() for x in eng_s.forward(seeds, walk_ids, **kw)]
        out_g = [x.clone() for x in eng_g.forward(seeds, walk_ids, **kw)]
        assert eng_s.err_flag.tolist() == [0, 0]
        for a, c in zip(eng_s.hops, eng_g.hops):
            assert torch.equal(a, c)
        for a, c in zip(out_s, out_g):
            close(a, c.double(), scale=float(c.abs().max()))
    # the ids also against the CPU walker
    want = walker.walk_vectorised(g.indptr, g.indices, seeds.cpu().numpy(), ns, rng_seed=77, first_walk=100)
    eng_s.forward(seeds, walk_ids, rng_seed=77)
    for a, c in zip(eng_s.hops, want):
        np.testing.assert_array_equal(a.cpu().numpy().astype(np.int64), c)


def test_subtree_route_with_isolated_and_invalid_seeds(cuda):
    """Seeds without neighbors and seeds outside the graph through the two-launch three-hop route: the same -1
    ids, error flags and embeddings as the sampler + layer kernels; hop-1 ids handed in give the same rows as
    hop-1 ids drawn by the kernel."""
    from graphsage_amd.engine import SageForwardEngine
    from graphsage_amd import kernels
    rng = np.random.default_rng(3)
    nu, ni, b = 120, 50, 11
    u = rng.integers(5, nu, 900)                                   # users 0..4 have no edges
    it = rng.integers(0, ni - 3, 900) + nu                         # the last three items neither
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, 64)).astype(np.float32),
             'item': rng.standard_normal((ni, 64)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    ns = [5, 4, 3]
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=4)
    model = ga.GraphSAGE([32, 32, 32], gen, seed=6).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.1, 0.1)
    su = [0, 3] + list(e[0, :b - 2])
    si = [nu + ni - 1] + list(e[1, :b - 1])
    seeds = torch.tensor(su + si, dtype=torch.int32, device=cuda)
    eng_s = SageForwardEngine(g, model, b, keep_ids=True, subtree_kernel=True)
    eng_g = SageForwardEngine(g, model, b, subtree_kernel=False)
    assert eng_s.subtree_kernel and not eng_g.subtree_kernel
    out_s = [x.clone() for x in eng_s.forward(seeds, None, rng_seed=9)]
    out_g = [x.clone() for x in eng_g.forward(seeds, None, rng_seed=9)]
    assert eng_s.err_flag.tolist() == eng_g.err_flag.tolist() == [nu + ni, 0]
    for a, c in zip(eng_s.hops, eng_g.hops):
        assert torch.equal(a, c)
    assert (eng_s.hops[0][0] == -1).all() and (eng_s.hops[2][b] == -1).all()
    for a, c in zip(out_s, out_g):
        close(a, c.double(), scale=float(c.abs().max()))
    # a seed outside the graph
    bad = seeds.clone()
    bad[4] = nu + ni + 7
    for eng in (eng_s, eng_g):
        eng.err_flag.zero_()
        eng.forward(bad, None, rng_seed=9)
    assert eng_s.err_flag.tolist()[1] == eng_g.err_flag.tolist()[1] == 1
    assert torch.equal(eng_s.hops[0], eng_g.hops[0]) and (eng_s.hops[0][4] == -1).all()
    # hop-1 ids handed in: the same rows, bit for bit
    st = eng_s._subtree
    eng_s.forward(seeds, None, rng_seed=9)
    h0, h1 = st['h0'].clone(), st['h1'].clone()
    m = model
    weights = {(l, t): (m._aggs[l][t].w_self, m._aggs[l][t].w_neigh, m._aggs[l][t].bias) for l in (0, 1) for t in ('user', 'item')}
    out_h0 = torch.empty_like(h0)
    call = kernels.Sage2Call(eng_s.indptr, eng_s.indices, eng_s.tables['user'][0], eng_s.tables['item'][0], ns[1:], weights,
                             False, b, subtrees={'roots_per_walk': ns[0], 'tape_len': sum(eng_s.sizes[1:]),
                                                 'tape_off_hop2': eng_s.sizes[1], 'tape_off_hop3': eng_s.sizes[1] + eng_s.sizes[2],
                                                 'out_h0': out_h0})
    err = torch.zeros(2, dtype=torch.int32, device=cuda)
    got = call.run(eng_s.hops[0].reshape(-1), err, rng_seed=9)
    assert torch.equal(got, h1) and torch.equal(out_h0, h0)


def test_forward_pipeline_gives_each_batch_the_embeddings_of_a_plain_forward(cuda):
    """Batches in flight on several streams (SageForwardEngine.pipeline): every batch's embeddings are bit for
    bit those of forward() on that batch, also when a slot is reused."""
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(21)
    nu, ni, b = 500, 120, 64
    u = np.concatenate([rng.integers(0, nu, 6000), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 6000), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, 64)).astype(np.float32),
             'item': rng.standard_normal((ni, 64)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, [8, 5], head_node_types=['user', 'item'], seed=4)
    model = ga.GraphSAGE([32, 32], gen, seed=6).build()
    eng = SageForwardEngine(g, model, b)
    assert eng.walk_kernel
    batches = [torch.tensor(list(e[0, k * b:(k + 1) * b]) + list(e[1, k * b:(k + 1) * b]), dtype=torch.int32, device=cuda)
               for k in range(7)]
    wid = [torch.arange(2 * b, dtype=torch.int64, device=cuda) + 1000 * k for k in range(7)]
    want = [[x.clone() for x in eng.forward(s, w, rng_seed=5)] for s, w in zip(batches, wid)]
    for depth in (1, 2, 3):
        pipe = eng.pipeline(depth)
        got = []
        for s, w in zip(batches, wid):
            eu, ei, ev = pipe.submit(s, w, rng_seed=5)
            ev.synchronize()
            got.append([eu.clone(), ei.clone()])
        pipe.drain()
        for a, c in zip(got, want):
            assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
        # without waiting per batch: the last `depth` batches are still in their slots afterwards
        outs = [pipe.submit(s, w, rng_seed=5) for s, w in zip(batches, wid)]
        pipe.drain()
        torch.cuda.synchronize()
        for k in range(len(batches) - depth, len(batches)):
            assert torch.equal(outs[k][0], want[k][0]) and torch.equal(outs[k][1], want[k][1])
    assert eng.err_flag.tolist() == [0, 0]
    # three hops: the two-launch route, every slot with its own hop-1 ids and activations
    three = ga.GraphSAGE([32, 32, 32], ga.GraphSAGEGenerator(g, b, [3, 3, 2], head_node_types=['user', 'item'], seed=4), seed=6).build()
    eng3 = SageForwardEngine(g, three, b)
    assert eng3.subtree_kernel
    want3 = [[x.clone() for x in eng3.forward(s, w, rng_seed=5)] for s, w in zip(batches, wid)]
    pipe = eng3.pipeline(2)
    outs = []
    for s, w in zip(batches, wid):
        eu, ei, ev = pipe.submit(s, w, rng_seed=5)
        ev.synchronize()
        outs.append([eu.clone(), ei.clone()])
    for a, c in zip(outs, want3):
        assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    from graphsage_amd._lib import GsageError
    with pytest.raises(GsageError):                                # the layered routes share their buffers
        SageForwardEngine(g, three, b, subtree_kernel=False).pipeline(2)


def test_forward_pipeline_on_padded_feature_widths_reads_the_current_weights(cuda):
    """Odd feature widths (28 / 18) run on zero-padded tables and padded copies of the layer-0 weights.  A
    pipeline slot refreshes its own copy on its own stream: the very first call may say inputs_ready, and
    weights changed between submits are the ones the next launch reads."""
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(22)
    nu, ni, b = 300, 90, 48
    u = np.concatenate([rng.integers(0, nu, 4000), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 4000), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, 28)).astype(np.float32),
             'item': rng.standard_normal((ni, 18)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, [6, 4], head_node_types=['user', 'item'], seed=4)
    model = ga.GraphSAGE([32, 32], gen, seed=6).build()
    batches = [torch.tensor(list(e[0, k * b:(k + 1) * b]) + list(e[1, k * b:(k + 1) * b]), dtype=torch.int32, device=cuda)
               for k in range(5)]
    wid = [torch.arange(2 * b, dtype=torch.int64, device=cuda) + 1000 * k for k in range(5)]
    eng = SageForwardEngine(g, model, b)
    assert eng.walk_kernel and eng._padded is not None
    pipe = eng.pipeline(2)
    torch.cuda.synchronize()
    # first call ever on this engine, inputs_ready: nothing on the current stream orders the weight copies
    first = pipe.submit(batches[0], wid[0], rng_seed=5, inputs_ready=True)
    first[2].synchronize()
    ref = SageForwardEngine(g, model, b)
    want = ref.forward(batches[0], wid[0], rng_seed=5)
    assert torch.equal(first[0], want[0]) and torch.equal(first[1], want[1])
    # the weights move between submits (a training loop scoring as it goes): each launch reads what is there
    for k in range(1, 5):
        with torch.no_grad():
            for t in ('user', 'item'):
                model._aggs[0][t].w_self.mul_(1.0 + 0.1 * k)
                model._aggs[0][t].w_neigh.add_(0.01 * k)
        out = pipe.submit(batches[k], wid[k], rng_seed=5)
        want = [x.clone() for x in ref.forward(batches[k], wid[k], rng_seed=5)]
        out[2].synchronize()
        assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])
    pipe.drain()
    assert eng.err_flag.tolist() == [0, 0]


def test_three_hop_fanouts_too_large_for_the_subtree_kernel_take_the_layered_route(cuda):
    """Fan-outs whose eight subtrees' ids do not fit a wavefront's LDS share: the engine asks
    gs_sage2_subtrees_supported, takes the sampler + one kernel per layer, and the results equal the general kernels'."""
    from graphsage_amd.engine import SageForwardEngine
    from graphsage_amd import kernels
    rng = np.random.default_rng(12)
    nu, ni, b = 200, 60, 4
    u = np.concatenate([rng.integers(0, nu, 2000), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 2000), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, 64)).astype(np.float32),
             'item': rng.standard_normal((ni, 64)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    ns = [2, 60, 60]
    assert kernels.sage2_supported(64, 64, 32, 32, 60, 60) and not kernels.sage2_subtrees_supported(64, 64, 32, 32, 60, 60)
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=4)
    model = ga.GraphSAGE([32, 32, 32], gen, seed=6).build()
    seeds = torch.tensor(list(e[0, :b]) + list(e[1, :b]), dtype=torch.int32, device=cuda)
    eng = SageForwardEngine(g, model, b)
    ref = SageForwardEngine(g, model, b, use_fused=False)
    assert not eng.subtree_kernel and all(eng.fused_layers)
    out = [x.clone() for x in eng.forward(seeds, None, rng_seed=2)]
    want = [x.clone() for x in ref.forward(seeds, None, rng_seed=2)]
    for a, c in zip(out, want):
        close(a, c.double(), scale=float(c.abs().max()))


def test_engine_wide_kernel_on_every_layer_matches_general_kernels(cuda):
    """A wide three-hop model with the row threshold lowered so that every layer -- dense activation
    rows from layer 1 on -- goes through the wide kernel: same embeddings as the general kernels."""
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(8)
    nu, ni, b = 300, 80, 40
    u = np.concatenate([rng.integers(0, nu, 2500), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 2500), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, 64)).astype(np.float32),
             'item': rng.standard_normal((ni, 128)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    ns, sizes = [6, 5, 3], [256, 256, 256]
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=2)
    model = ga.GraphSAGE(sizes, gen, seed=9).build()
    seeds = torch.tensor(list(e[0, :b]) + list(e[1, :b]), dtype=torch.int32, device=cuda)
    walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda)
    eng_w = SageForwardEngine(g, model, b, wide_min_rows=1)
    eng_d = SageForwardEngine(g, model, b)                     # default threshold: wide layer 0, general tail
    eng_u = SageForwardEngine(g, model, b, use_fused=False)
    assert eng_w.wide_layers == [True, True, True] and eng_d.wide_layers == [True, False, False]
    out_w = [x.clone() for x in eng_w.forward(seeds, walk_ids, rng_seed=5)]
    out_d = [x.clone() for x in eng_d.forward(seeds, walk_ids, rng_seed=5)]
    out_u = [x.clone() for x in eng_u.forward(seeds, walk_ids, rng_seed=5)]
    for a, c, d in zip(out_w, out_u, out_d):
        assert bool(torch.isfinite(a).all())
        close(a, c.double(), scale=float(c.abs().max()))
        close(d, c.double(), scale=float(c.abs().max()))


@pytest.mark.parametrize('ns,sizes,fu,fi', [([8, 4], [32, 32], 64, 64), ([25, 10], [32, 32], 64, 32),
                                            ([4, 3, 2], [32, 64, 32], 32, 64), ([8, 4], [32, 32], 24, 19),
                                            ([5, 3, 2], [256, 256, 256], 64, 128)])     # wide layers: wide kernel
def test_engine_fused_and_unfused_match_model_and_oracle(cuda, ns, sizes, fu, fi):
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(3)
    nu, ni, b = 200, 50, 24
    u = np.concatenate([rng.integers(0, nu, 1500), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 1500), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, fu)).astype(np.float32),
             'item': rng.standard_normal((ni, fi)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=21)
    model = ga.GraphSAGE(sizes, gen, seed=5).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.1, 0.1)
    links = [(int(e[0, i]), int(e[1, i])) for i in range(b)]
    seeds = torch.tensor([l[0] for l in links] + [l[1] for l in links], dtype=torch.int32, device=cuda)
    walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda)

    # keep_ids: the layer-0 kernel draws the deepest hop itself and also leaves its ids in hops[-1]
    eng_f = SageForwardEngine(g, model, b, use_fused=True, walk_kernel=False, keep_ids=True, draw_in_layer=True)
    eng_u = SageForwardEngine(g, model, b, use_fused=False)
    fused_shapes = fu % 32 == 0 and fi % 32 == 0
    assert all(eng_f.fused_layers) == fused_shapes
    out_f = [x.clone() for x in eng_f.forward(seeds, walk_ids, rng_seed=99)]
    out_u = [x.clone() for x in eng_u.forward(seeds, walk_ids, rng_seed=99)]
    assert eng_f.err_flag.tolist() == [0, 0]
    for a, c in zip(eng_f.hops, eng_u.hops):
        assert torch.equal(a, c)

    # the oracle on the very same sampled ids
    hops = [h.cpu().numpy().astype(np.int64) for h in eng_f.hops]
    want_hops = walker.walk_vectorised(g.indptr, g.indices, seeds.cpu().numpy(), ns, rng_seed=99, first_walk=0)
    for a, c in zip(hops, want_hops):
        np.testing.assert_array_equal(a, c)
    xin = ogen.sample_features_fast({k: v.astype(np.float64) for k, v in feats.items()}, nu, links, hops)
    plan = omodel.LayerPlan(sizes, ns, g.node_feature_sizes)
    want = omodel.forward(plan, _oracle_model(model, plan), [torch.from_numpy(x) for x in xin])
    for k in range(2):
        close(out_f[k], want[k], scale=1.0)
        close(out_u[k], want[k], scale=1.0)
    # second call on other seeds reuses the prepared launches
    seeds2 = seeds.flip(0).contiguous()
    seeds2 = torch.cat([seeds2[b:], seeds2[:b]])
    o1 = [x.clone() for x in eng_f.forward(seeds2, walk_ids, rng_seed=5)]
    o2 = [x.clone() for x in eng_u.forward(seeds2, walk_ids, rng_seed=5)]
    for k in range(2):
        close(o1[k], o2[k], scale=1.0)
    assert not torch.equal(o1[0], out_f[0])


# -------------------------------------------- whole forward in one kernel ----

@pytest.mark.parametrize('case', [0, 2, 3, 5])
def test_walk_kernel_replay_against_reference_batches(golden, cuda, case):
    """gs_sage2_forward on replayed draws: sampled ids equal the reference's walks
    bit for bit, embeddings equal the float64 oracle applied to the reference
    generator's own feature arrays (graph features padded to a supported width)."""
    arrays, meta = golden
    info = meta['replay'][case]
    key, ns, b = info['key'], info['num_samples'], info['batch']
    feats, nodes, edges = golden_graph_parts(arrays, info['graph'])
    pad = lambda a, f: np.concatenate([a, np.zeros((a.shape[0], f - a.shape[1]), np.float32)], axis=1)
    feats = {'user': pad(feats['user'], 32), 'item': pad(feats['item'], 32)}
    g = ga.Graph(feats, nodes, edges)
    links = arrays[key + '_links']
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=1)
    model = ga.GraphSAGE([32, 32], gen, seed=13).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.1, 0.1)
    dev = g.device_tensors()
    weights = {(l, t): (model._aggs[l][t].w_self, model._aggs[l][t].w_neigh, model._aggs[l][t].bias)
               for l in (0, 1) for t in ('user', 'item')}
    call = kernels.Sage2Call(dev['indptr'], dev['indices'], dev['user'], dev['item'], ns, weights, True, b,
                             keep_ids=True)
    seeds = torch.from_numpy(np.concatenate([links[:, 0], links[:, 1]]).astype(np.int32)).to(cuda)
    tape = torch.from_numpy(arrays[key + '_tape'].view(np.int32)).to(cuda)
    err = torch.zeros(2, dtype=torch.int32, device=cuda)
    emb = call.run(seeds, err, replay=tape)
    assert err.tolist() == [0, 0]
    np.testing.assert_array_equal(call.hop1.cpu().numpy(), arrays[key + '_hop1'])
    np.testing.assert_array_equal(call.hop2.cpu().numpy(), arrays[key + '_hop2'])
    xin = [torch.from_numpy(pad(arrays[key + '_x%d' % i].reshape(-1, arrays[key + '_x%d' % i].shape[2]), 32)
                            .reshape(b, -1, 32)).double() for i in range(6)]
    plan = omodel.LayerPlan([32, 32], ns, g.node_feature_sizes)
    want = omodel.forward(plan, _oracle_model(model, plan), xin)
    close(emb[:b], want[0], scale=1.0)
    close(emb[b:], want[1], scale=1.0)


@pytest.mark.parametrize('case', [0, 1, 2, 3, 4, 5, 6])
def test_layer_kernel_draws_replay_against_reference_batches(golden, cuda, case):
    """gs_sample over all hops but the last + gs_sage_fused_sampled_fwd drawing the last hop inside the
    layer-0 kernel, on the reference's replayed tapes: the ids the kernel drew equal the reference's
    walks bit for bit (its hop-2 lists), embeddings equal the float64 oracle applied to the reference
    generator's own feature arrays."""
    from graphsage_amd.engine import SageForwardEngine
    arrays, meta = golden
    info = meta['replay'][case]
    key, ns, b = info['key'], info['num_samples'], info['batch']
    feats, nodes, edges = golden_graph_parts(arrays, info['graph'])
    pad = lambda a, f: np.concatenate([a, np.zeros((a.shape[0], f - a.shape[1]), np.float32)], axis=1)
    feats = {'user': pad(feats['user'], 32), 'item': pad(feats['item'], 32)}
    g = ga.Graph(feats, nodes, edges)
    links = arrays[key + '_links']
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=1)
    model = ga.GraphSAGE([32, 32], gen, seed=13).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.1, 0.1)
    eng = SageForwardEngine(g, model, b, walk_kernel=False, keep_ids=True, draw_in_layer=True)
    assert eng.draw_deepest and all(eng.fused_layers)
    seeds = torch.from_numpy(np.concatenate([links[:, 0], links[:, 1]]).astype(np.int32)).to(cuda)
    tape = torch.from_numpy(arrays[key + '_tape'].view(np.int32)).to(cuda)
    eng.hops[1].fill_(-5)                                          # the sampler kernel must not be what fills it
    emb = eng.forward(seeds, replay=tape)
    assert eng.err_flag.tolist() == [0, 0]
    np.testing.assert_array_equal(eng.hops[0].cpu().numpy(), arrays[key + '_hop1'])
    np.testing.assert_array_equal(eng.hops[1].cpu().numpy(), arrays[key + '_hop2'])
    xin = [torch.from_numpy(pad(arrays[key + '_x%d' % i].reshape(-1, arrays[key + '_x%d' % i].shape[2]), 32)
                            .reshape(b, -1, 32)).double() for i in range(6)]
    plan = omodel.LayerPlan([32, 32], ns, g.node_feature_sizes)
    want = omodel.forward(plan, _oracle_model(model, plan), xin)
    close(emb[0], want[0], scale=1.0)
    close(emb[1], want[1], scale=1.0)
    # without keep_ids nothing is written there and the embeddings are the same bits
    eng2 = SageForwardEngine(g, model, b, walk_kernel=False, draw_in_layer=True)
    eng2.hops[1].fill_(-5)
    emb2 = eng2.forward(seeds, replay=tape)
    assert torch.equal(emb2[0], emb[0]) and torch.equal(emb2[1], emb[1])
    assert bool((eng2.hops[1] == -5).all())
    # and the sampler kernel drawing every hop gives the same embeddings
    eng3 = SageForwardEngine(g, model, b, walk_kernel=False, draw_in_layer=False)
    assert not eng3.draw_deepest
    emb3 = eng3.forward(seeds, replay=tape)
    assert torch.equal(emb3[0], emb[0]) and torch.equal(emb3[1], emb[1])
    np.testing.assert_array_equal(eng3.hops[1].cpu().numpy(), arrays[key + '_hop2'])


@pytest.mark.parametrize('ns', [[4, 3, 2], [7], [3, 2, 2, 2], [25, 10, 5]])
def test_layer_kernel_draws_equal_the_sampler_kernel_at_any_depth(golden, cuda, ns):
    """Philox mode, depths 1 to 4, walk ids beyond 32 bits: the ids drawn inside the layer-0 kernel are
    the ones gs_sample draws for that hop, and a node without neighbors at the last-but-one hop is
    reported through err_flag with -1 children, as gs_sample does."""
    from graphsage_amd.engine import SageForwardEngine
    arrays, _ = golden
    feats, nodes, edges = golden_graph_parts(arrays, 'g_mid')
    pad = lambda a, f: np.concatenate([a, np.zeros((a.shape[0], f - a.shape[1]), np.float32)], axis=1)
    feats = {'user': pad(feats['user'], 32), 'item': pad(feats['item'], 32)}
    g = ga.Graph(feats, nodes, edges)
    b = 19
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=1)
    model = ga.GraphSAGE([32] * len(ns), gen, seed=13).build()
    nu = feats['user'].shape[0]
    half = edges.shape[1] // 2
    seeds = torch.from_numpy(np.concatenate([edges[0, :b], edges[1, :b]]).astype(np.int32)).to(cuda)
    assert int(seeds[:b].max()) < nu <= int(seeds[b:].min()) and half >= b
    walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda) * 5 + (1 << 40)
    eng = SageForwardEngine(g, model, b, keep_ids=True, walk_kernel=False, draw_in_layer=True)
    assert eng.draw_deepest
    emb = [x.clone() for x in eng.forward(seeds, walk_ids, rng_seed=0xABCDEF0123)]
    dev = g.device_tensors()
    hops, err = kernels.sample(dev['indptr'], dev['indices'], seeds, ns, rng_seed=0xABCDEF0123, walk_ids=walk_ids)
    assert err.tolist() == [0, 0] and eng.err_flag.tolist() == [0, 0]
    for a, c in zip(eng.hops, hops):
        assert torch.equal(a, c)
    ref = SageForwardEngine(g, model, b, use_fused=False)
    out = ref.forward(seeds, walk_ids, rng_seed=0xABCDEF0123)
    for k in range(2):
        close(emb[k], out[k], scale=1.0)
    if len(ns) >= 2:
        # cut one hop-(L-2) node's adjacency row to nothing: an isolated parent at the drawn hop
        indptr = dev['indptr'].clone()
        victim = int(eng.hops[-2][0, 0])
        lo, hi = int(indptr[victim]), int(indptr[victim + 1])
        indptr[victim + 1:] -= (hi - lo)
        indices = torch.cat([dev['indices'][:lo], dev['indices'][hi:]]).contiguous()
        g2 = ga.Graph.from_csr({'user': dev['user'], 'item': dev['item']}, indptr, indices)
        eng_i = SageForwardEngine(g2, model, b, keep_ids=True, walk_kernel=False, draw_in_layer=True)
        eng_i.forward(seeds, walk_ids, rng_seed=0xABCDEF0123)
        hops_i, err_i = kernels.sample(indptr, indices, seeds, ns, rng_seed=0xABCDEF0123, walk_ids=walk_ids)
        assert err_i.tolist()[0] >= 1 and eng_i.err_flag.tolist() == err_i.tolist()
        for a, c in zip(eng_i.hops, hops_i):
            assert torch.equal(a, c)
        assert bool((eng_i.hops[-1] == -1).any())


@pytest.mark.parametrize('ns,sizes,fu,fi', [([25, 10], [32, 32], 64, 64), ([8, 4], [64, 64], 64, 64),
                                            ([10, 5], [32, 32], 64, 32), ([33, 7], [32, 32], 128, 128),
                                            ([64, 3], [32, 64], 64, 64), ([3, 70], [32, 32], 32, 32),
                                            ([10, 5], [32, 32], 24, 19), ([8, 4], [32, 32], 28, 18)])
def test_walk_kernel_matches_layered_engine_and_oracle(cuda, ns, sizes, fu, fi):
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(7)
    nu, ni, b = 300, 60, 37
    u = np.concatenate([rng.integers(0, nu, 2500), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 2500), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, fu)).astype(np.float32),
             'item': rng.standard_normal((ni, fi)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=2)
    model = ga.GraphSAGE(sizes, gen, seed=3).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.1, 0.1)
    links = [(int(e[0, i]), int(e[1, i])) for i in range(b)]
    seeds = torch.tensor([l[0] for l in links] + [l[1] for l in links], dtype=torch.int32, device=cuda)
    walk_ids = (torch.arange(2 * b, dtype=torch.int64, device=cuda) * 3 + (1 << 35))
    eng_w = SageForwardEngine(g, model, b)
    eng_l = SageForwardEngine(g, model, b, walk_kernel=False, keep_ids=True, draw_in_layer=True)
    assert eng_w.walk_kernel and not eng_l.walk_kernel
    out_w = [x.clone() for x in eng_w.forward(seeds, walk_ids, rng_seed=4242)]
    out_l = [x.clone() for x in eng_l.forward(seeds, walk_ids, rng_seed=4242)]
    assert eng_w.err_flag.tolist() == [0, 0]
    hops = [h.cpu().numpy().astype(np.int64) for h in eng_l.hops]
    xin = ogen.sample_features_fast({k: v.astype(np.float64) for k, v in feats.items()}, nu, links, hops)
    plan = omodel.LayerPlan(sizes, ns, g.node_feature_sizes)
    want = omodel.forward(plan, _oracle_model(model, plan), [torch.from_numpy(x) for x in xin])
    for k in range(2):
        close(out_w[k], want[k], scale=1.0)
        close(out_l[k], want[k], scale=1.0)
    # isolated node / bad seed handling inside the one-kernel path
    bad = seeds.clone()
    bad[3] = nu + ni + 5
    eng_w.err_flag.zero_()
    eng_w.forward(bad, walk_ids, rng_seed=1)
    assert eng_w.err_flag.tolist()[1] == 1
    if fu % 32:
        # odd widths run on zero-padded copies of the tables and of the layer-0 weights; a weight
        # change made behind torch's back (as the Adam kernel does) must still reach the copies
        assert eng_w._padded is not None
        agg = model._aggs[0]['user']
        with torch.no_grad():
            agg.w_self.mul_(0.5)
        out2 = [x.clone() for x in eng_w.forward(seeds, walk_ids, rng_seed=4242)]
        out3 = [x.clone() for x in eng_l.forward(seeds, walk_ids, rng_seed=4242)]
        assert not torch.equal(out2[0], out_w[0])
        for k in range(2):
            close(out2[k], out3[k], scale=1.0)


def test_engine_runs_the_max_aggregator_through_the_general_kernels(cuda):
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(11)
    nu, ni, b, ns = 120, 30, 16, [6, 3]
    u = np.concatenate([rng.integers(0, nu, 900), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 900), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, 64)).astype(np.float32),
             'item': rng.standard_normal((ni, 64)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=2)
    model = ga.GraphSAGE([32, 32], gen, agg_option='max', seed=3).build()
    links = [(int(e[0, i]), int(e[1, i])) for i in range(b)]
    seeds = torch.tensor([l[0] for l in links] + [l[1] for l in links], dtype=torch.int32, device=cuda)
    walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda)
    eng = SageForwardEngine(g, model, b)
    assert eng.agg_max and not eng.walk_kernel and not any(eng.fused_layers)
    out = [x.clone() for x in eng.forward(seeds, walk_ids, rng_seed=8)]
    hops = [h.cpu().numpy().astype(np.int64) for h in eng.hops]
    xin = ogen.sample_features_fast({k: v.astype(np.float64) for k, v in feats.items()}, nu, links, hops)
    plan = omodel.LayerPlan([32, 32], ns, g.node_feature_sizes)
    want = omodel.forward(plan, _oracle_model(model, plan), [torch.from_numpy(x) for x in xin], agg='max')
    for k in range(2):
        close(out[k], want[k], scale=1.0)
    from graphsage_amd.train_engine import SageTrainEngine
    with pytest.raises(Exception):
        SageTrainEngine(g, model, b)


@pytest.mark.parametrize('m,fs,fn,h', [(300, 64, 64, 32), (77, 24, 19, 32), (700, 256, 256, 256), (1100, 64, 128, 256)])
def test_projection_with_own_rows_gathered_in_the_kernel(cuda, m, fs, fn, h):
    """gs_sage_project_gather_fwd == gs_gather_rows followed by gs_sage_project_fwd,
    bit for bit (same kernel arithmetic), including ids outside the table (zero rows)."""
    rng = np.random.default_rng(m + fs)
    n_rows, off = 5000, 1234
    feat = torch.from_numpy(rng.standard_normal((n_rows, fs)).astype(np.float32)).to(cuda)
    ids = rng.integers(0, n_rows, m).astype(np.int32) + off
    ids[3], ids[m - 1] = off - 1, off + n_rows                       # just outside, both ends
    ids_t = torch.from_numpy(ids).to(cuda)
    x_mean = torch.from_numpy(rng.standard_normal((m, fn)).astype(np.float32)).to(cuda)
    ws = torch.from_numpy((rng.standard_normal((fs, h // 2)) * 0.1).astype(np.float32)).to(cuda)
    wn = torch.from_numpy((rng.standard_normal((fn, h // 2)) * 0.1).astype(np.float32)).to(cuda)
    b = torch.from_numpy(rng.standard_normal(h).astype(np.float32)).to(cuda)
    for act in ('relu', 'linear'):
        got = kernels.sage_project_gather(feat, ids_t, off, x_mean, ws, wn, b, act)
        x_self = kernels.gather_rows(feat, ids_t, off)
        assert float(x_self[3].abs().max()) == 0.0 and float(x_self[m - 1].abs().max()) == 0.0
        want = kernels.sage_project(x_self, x_mean, ws, wn, b, act)
        assert torch.equal(got, want)
    with pytest.raises(ValueError):
        kernels.sage_project_gather(feat, ids_t[:-1].contiguous(), off, x_mean, ws, wn, b, 'relu')


def test_launch_timer_brackets_exactly_one_kernel(cuda):
    """gs_timer_*: the armed pair rides on the next gather-mean launch only."""
    feat = torch.randn(200_000, 64, device=cuda)
    ids = torch.randint(0, 200_000, (50_000 * 10,), device=cuda, dtype=torch.int32)
    t = kernels.LaunchTimer()
    t.arm()
    a = kernels.gather_mean(feat, ids, 50_000, 10)
    b = kernels.gather_mean(feat, ids, 50_000, 10)            # not timed: the timer disarmed itself
    us = t.elapsed_us()
    assert 1.0 < us < 5_000.0, us
    assert torch.equal(a, b)
    t2 = kernels.LaunchTimer()
    t2.arm()
    kernels.gather_rows(feat, ids[:1000].contiguous())        # not one of the timed kernels: stays armed
    kernels.gather_mean(feat, ids, 50_000, 10)
    assert t2.elapsed_us() > 1.0


@pytest.mark.parametrize('ns,sizes,f', [([8, 4], [32, 32], 64), ([4, 3, 2], [32, 64, 32], 32)])
def test_engines_without_bias_and_without_normalisation(cuda, ns, sizes, f):
    """GraphSAGE(bias=False, normalize=None) (models.py:19-35): every forward route agrees with the oracle."""
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(8)
    nu, ni, b = 200, 50, 24
    u = np.concatenate([rng.integers(0, nu, 1500), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 1500), np.arange(nu) % ni]) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, f)).astype(np.float32),
             'item': rng.standard_normal((ni, f)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=21)
    model = ga.GraphSAGE(sizes, gen, bias=False, normalize=None, seed=5).build()
    assert all(a.bias is None for aggs in model._aggs for a in aggs.values())
    links = [(int(e[0, i]), int(e[1, i])) for i in range(b)]
    seeds = torch.tensor([l[0] for l in links] + [l[1] for l in links], dtype=torch.int32, device=cuda)
    walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda)
    engines = [SageForwardEngine(g, model, b), SageForwardEngine(g, model, b, walk_kernel=False),
               SageForwardEngine(g, model, b, use_fused=False)]
    outs = [[x.clone() for x in eng.forward(seeds, walk_ids, rng_seed=5)] for eng in engines]
    want_hops = walker.walk_vectorised(g.indptr, g.indices, seeds.cpu().numpy(), ns, rng_seed=5, first_walk=0)
    xin = ogen.sample_features_fast({k: v.astype(np.float64) for k, v in feats.items()}, nu, links, want_hops)
    plan = omodel.LayerPlan(sizes, ns, g.node_feature_sizes)
    want = omodel.forward(plan, _oracle_model(model, plan), [torch.from_numpy(x) for x in xin], normalize=None)
    for out in outs:
        for k in range(2):
            close(out[k], want[k], scale=float(want[k].abs().max()))


# ----------------------------- the reference's cyclic walker (backup/sampler_backup.py) ----

def _cyclic_perms(arrays, key):
    lengths, flat, out, pos = arrays[key + '_perm_lengths'], arrays[key + '_perms'], [], 0
    for n in lengths:
        out.append(flat[pos:pos + int(n)])
        pos += int(n)
    return out


@pytest.mark.parametrize('case', range(4))
def test_cyclic_walker_replay_is_bit_exact_with_the_reference(golden, cuda, case):
    """gs_sample_cyclic fed the permutations the reference's backup/sampler_backup.py was given: the same
    walks call after call (the cursors carry over), the same cursors at the end, the same nested lists."""
    from graphsage_amd import sampler_backup
    arrays, meta = golden
    info = meta['cyclic'][case]
    key, ns = info['key'], info['num_samples']
    g = ga.Graph(*golden_graph_parts(arrays, info['graph']))
    w = sampler_backup.BreadthFirstWalker(g, ns)
    w.set_replay(_cyclic_perms(arrays, key))
    nested = []
    for k, roots in enumerate(info['calls']):
        walks = w.run_breadth_first_walk(roots)
        nested.append(walks)
        for h, hop in enumerate(walker.walks_to_hops(walks, ns)):
            np.testing.assert_array_equal(hop, arrays['%s_call%d_hop%d' % (key, k, h + 1)])
    np.testing.assert_array_equal(g.sampling_cursor().cpu().numpy(), arrays[key + '_cursor'])
    assert int(w._state[0]) == info['n_perms'] and int(w._state[1]) == arrays[key + '_perms'].shape[0]
    if case == 0:
        assert nested == meta['cyclic0_nested']
        assert g.sampling_index_dict == {v: int(c) for v, c in enumerate(arrays[key + '_cursor'])}


@pytest.mark.parametrize('ns', [[3, 2], [5, 4, 3], [70]])
def test_cyclic_walker_with_its_own_permutations_equals_the_oracle(golden, cuda, ns):
    """Without a tape the permutations come from the Philox-keyed Feistel network; the oracle's statement of
    the walk with the same generator gives the same ids and cursors, over two calls."""
    from graphsage_amd import sampler_backup
    from oracle import cyclic
    from oracle.graph import OracleGraph
    arrays, _ = golden
    parts = golden_graph_parts(arrays, 'g_small')
    g, og = ga.Graph(*parts), OracleGraph(*parts)
    n_all = g.num_nodes
    w = sampler_backup.BreadthFirstWalker(g, ns, seed=0xFEEDC0FFEE)
    cursor = {v: 0 for v in range(n_all)}
    perm = cyclic.FeistelPerm(0xFEEDC0FFEE)
    for roots in ([39, 51, 0, 0, 45], [45, 2, 39]):
        got = w.run_breadth_first_walk(roots)
        want = cyclic.walk_cyclic(og, ns, roots, cursor, perm)
        assert got == want
    np.testing.assert_array_equal(g.sampling_cursor().cpu().numpy(), [cursor[v] for v in range(n_all)])
    assert int(w._state[0]) == perm.t


def test_cyclic_walker_errors(golden, cuda):
    from graphsage_amd import sampler_backup
    arrays, _ = golden
    g = ga.Graph(*golden_graph_parts(arrays, 'g_small'))
    with pytest.raises(TypeError):
        sampler_backup.BreadthFirstWalker(g, (3, 2))
    w = sampler_backup.BreadthFirstWalker(g, [8, 4])
    with pytest.raises(KeyError):
        w.run_breadth_first_walk([10 ** 6])
    assert w.run_breadth_first_walk([]) == []
    w.set_replay([])                                              # user 39 has one neighbor: the first visit wraps at once
    with pytest.raises(ValueError):
        w.run_breadth_first_walk([39])
    # a node without neighbors: the reference raises ValueError (sampler_backup.py:100)
    edges = np.array([[0, 1, 3, 3], [3, 3, 0, 1]])
    g2 = ga.Graph({'user': np.zeros((3, 2), np.float32), 'item': np.zeros((2, 2), np.float32)},
                  {'user': [0, 1, 2], 'item': [3, 4]}, edges)
    g2.indptr                                                     # (user 2 and item 4 have no outgoing edge)
    with pytest.raises(ValueError):
        sampler_backup.BreadthFirstWalker(g2, [2, 2]).sample_device(torch.tensor([2], dtype=torch.int32, device=cuda))
