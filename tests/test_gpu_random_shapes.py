"""Seeded sweeps over odd shapes: every kernel entry point against numpy / the
oracle on sizes that hit its tail paths (widths that are not multiples of the
vector width, fan-outs above the wave width, single rows, many children) --
needs a GPU."""
import numpy as np
import pytest
import torch

from graphsage_amd import kernels
from oracle import walker

pytestmark = pytest.mark.gpu


def _graph(rng, n_nodes, n_edges):
    src = np.concatenate([np.arange(n_nodes), rng.integers(0, n_nodes, n_edges)]).astype(np.int32)
    dst = rng.integers(0, n_nodes, src.size).astype(np.int32)
    order = np.argsort(src, kind='stable')
    indptr = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n_nodes))]).astype(np.int64)
    return src, dst, indptr, dst[order]


@pytest.mark.parametrize('seed', range(12))
def test_sampler_random_fanouts_against_the_oracle(cuda, seed):
    rng = np.random.default_rng(seed)
    n_nodes = int(rng.integers(5, 400))
    src, dst, indptr, indices = _graph(rng, n_nodes, int(rng.integers(0, 3000)))
    n_hops = int(rng.integers(1, 5))
    fan = [int(rng.integers(1, [300, 40, 6, 3][h] + 1)) for h in range(n_hops)]
    while np.prod(fan) > 20000:
        fan[int(np.argmax(fan))] //= 2
    roots = rng.integers(0, n_nodes, int(rng.integers(1, 70))).astype(np.int32)
    key, base = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 40))
    a, b = torch.from_numpy(src).to(cuda), torch.from_numpy(dst).to(cuda)
    d_indptr, d_indices = kernels.build_csr(a, b, n_nodes, sort_chunk=int(rng.choice([0, 17, 1000])))
    np.testing.assert_array_equal(d_indptr.cpu().numpy(), indptr)
    np.testing.assert_array_equal(d_indices.cpu().numpy(), indices)
    hops, err = kernels.sample(d_indptr, d_indices, torch.from_numpy(roots).to(cuda), fan, rng_seed=key, walk_id_base=base)
    assert err.tolist() == [0, 0]
    want = walker.walk_vectorised(indptr, indices, roots, fan, rng_seed=key, first_walk=base)
    for got, w in zip(hops, want):
        np.testing.assert_array_equal(got.cpu().numpy(), w)
    # the same draws from a tape
    sizes = walker.hop_sizes(fan)
    tape = rng.integers(0, 2 ** 32, len(roots) * sum(sizes), dtype=np.uint64).astype(np.uint32)
    hops, _ = kernels.sample(d_indptr, d_indices, torch.from_numpy(roots).to(cuda), fan,
                             replay=torch.from_numpy(tape.view(np.int32)).to(cuda))
    want = walker.walk_vectorised(indptr, indices, roots, fan, tape=tape)
    for got, w in zip(hops, want):
        np.testing.assert_array_equal(got.cpu().numpy(), w)


@pytest.mark.parametrize('seed', range(16))
def test_gather_and_reduce_random_shapes(cuda, seed):
    rng = np.random.default_rng(100 + seed)
    f = int(rng.choice([1, 2, 3, 5, 7, 12, 18, 24, 28, 33, 64, 100, 130, 256, 514]))
    n_rows = int(rng.integers(1, 3000))
    n_child = int(rng.choice([1, 2, 3, 9, 25, 64, 130, 700]))
    n_parents = int(rng.integers(1, 200 if n_child < 100 else 20))
    off = int(rng.integers(0, 1000))
    feat = rng.standard_normal((n_rows, f)).astype(np.float32)
    ids = rng.integers(0, n_rows, (n_parents, n_child)).astype(np.int32)
    t, d_ids = torch.from_numpy(feat).to(cuda), torch.from_numpy(ids + off).to(cuda).reshape(-1)
    rows = kernels.gather_rows(t, d_ids, off)
    np.testing.assert_array_equal(rows.cpu().numpy(), feat[ids.reshape(-1)])
    mean = kernels.gather_mean(t, d_ids, n_parents, n_child, off).cpu().numpy()
    want = feat[ids].astype(np.float64).mean(axis=1)
    assert np.abs(mean - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    mx = kernels.gather_max(t, d_ids, n_parents, n_child, off).cpu().numpy()
    np.testing.assert_array_equal(mx, feat[ids].max(axis=1))
    # dense groups and their gradients
    m = (n_rows // n_child)
    if m:
        dense = t[:m * n_child].contiguous()
        got = kernels.gather_mean(dense, None, m, n_child).cpu().numpy()
        want = feat[:m * n_child].reshape(m, n_child, f).astype(np.float64).mean(axis=1)
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
        g = torch.from_numpy(rng.standard_normal((m, f)).astype(np.float32)).to(cuda)
        gx = torch.empty_like(dense)
        kernels.gather_mean_bwd(g, None, n_child, gx)
        np.testing.assert_allclose(gx.cpu().numpy().reshape(m, n_child, f),
                                   np.repeat((g.cpu().numpy() / np.float32(n_child))[:, None, :], n_child, axis=1),
                                   rtol=1e-6, atol=0)


@pytest.mark.parametrize('seed', range(16))
def test_projection_random_shapes_forward_and_backward(cuda, seed):
    rng = np.random.default_rng(200 + seed)
    m = int(rng.choice([1, 3, 63, 64, 65, 127, 500, 513, 1030]))
    fs, fn = int(rng.choice([1, 5, 19, 24, 32, 64, 96, 128, 160])), int(rng.choice([1, 7, 18, 28, 32, 64, 128]))
    h = 2 * int(rng.choice([1, 3, 8, 16, 20, 32, 64, 128]))
    act = str(rng.choice(['relu', 'linear']))
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(cuda)
    xs, xm = rng.standard_normal((m, fs)), rng.standard_normal((m, fn))
    ws, wn, b = 0.3 * rng.standard_normal((fs, h // 2)), 0.3 * rng.standard_normal((fn, h // 2)), rng.standard_normal(h)
    out = kernels.sage_project(f32(xs), f32(xm), f32(ws), f32(wn), f32(b), act)
    z = np.concatenate([xs.astype(np.float32).astype(np.float64) @ ws.astype(np.float32).astype(np.float64),
                        xm.astype(np.float32).astype(np.float64) @ wn.astype(np.float32).astype(np.float64)], axis=1)
    z = z + b.astype(np.float32).astype(np.float64)
    want = np.maximum(z, 0) if act == 'relu' else z
    scale = max(1.0, np.abs(want).max())
    assert np.abs(out.cpu().numpy() - want).max() <= 2e-5 * scale
    # backward against the same formulas
    g = rng.standard_normal((m, h)).astype(np.float32)
    res = dict(zip(('g_w_self', 'g_w_neigh', 'g_bias', 'g_x_self', 'g_x_mean'),
                   kernels.sage_project_bwd(f32(xs), f32(xm), f32(ws), f32(wn), act, out, f32(g))))
    gm = g.astype(np.float64) * ((want > 0) if act == 'relu' else 1.0)
    x32s, x32m = xs.astype(np.float32).astype(np.float64), xm.astype(np.float32).astype(np.float64)
    wants = {'g_w_self': x32s.T @ gm[:, :h // 2], 'g_w_neigh': x32m.T @ gm[:, h // 2:], 'g_bias': gm.sum(0),
             'g_x_self': gm[:, :h // 2] @ ws.astype(np.float32).astype(np.float64).T,
             'g_x_mean': gm[:, h // 2:] @ wn.astype(np.float32).astype(np.float64).T}
    for name, w in wants.items():
        got = res[name].cpu().numpy()
        assert got.shape == w.shape, name
        assert np.abs(got - w).max() <= 3e-5 * max(1.0, np.abs(w).max()), name


def _check_engine_against_oracle(cuda, rng, seed, fan, fu, fi, sizes, agg, nu, ni, b):
    """Embeddings of the forward engine equal the float64 oracle applied to the ids the kernels sampled."""
    import graphsage_amd as ga
    from graphsage_amd.engine import SageForwardEngine
    from oracle import generator as ogen
    from oracle import model as omodel
    n_hops = len(fan)
    u = np.concatenate([rng.integers(0, nu, 1500), np.arange(nu)])
    it = np.concatenate([rng.integers(0, ni, 1500), np.arange(nu) % ni]) + nu
    for extra in range(ni):                                        # every item gets a neighbor
        u, it = np.append(u, extra % nu), np.append(it, nu + extra)
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, fu)).astype(np.float32),
             'item': rng.standard_normal((ni, fi)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))}, np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, fan, head_node_types=['user', 'item'], seed=seed)
    model = ga.GraphSAGE(sizes, gen, agg_option=agg, seed=seed + 1).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.2, 0.2)
    pick = rng.integers(0, e.shape[1], b)
    links = [(int(e[0, i]), int(e[1, i])) for i in pick]
    seeds = torch.tensor([l[0] for l in links] + [l[1] for l in links], dtype=torch.int32, device=cuda)
    walk_ids = torch.from_numpy(rng.integers(0, 2 ** 40, 2 * b)).to(cuda)
    eng = SageForwardEngine(g, model, b)
    out = [x.clone() for x in eng.forward(seeds, walk_ids, rng_seed=seed * 7 + 1)]
    assert eng.err_flag.tolist() == [0, 0]
    # the draws are a function of (key, walk id, hop, slot): the oracle reproduces them walk by walk
    hops = [np.empty((2 * b, s), dtype=np.int64) for s in walker.hop_sizes(fan)]
    wid = walk_ids.cpu().numpy()
    for w in range(2 * b):
        one = walker.walk_vectorised(g.indptr, g.indices, seeds[w:w + 1].cpu().numpy(), fan, rng_seed=seed * 7 + 1,
                                     first_walk=int(wid[w]))
        for h in range(n_hops):
            hops[h][w] = one[h][0]
    xin = ogen.sample_features_fast({k: v.astype(np.float64) for k, v in feats.items()}, nu, links, hops)
    plan = omodel.LayerPlan(sizes, fan, g.node_feature_sizes)
    weights = {(l, t): {'w_self': a.w_self.detach().cpu().double(), 'w_neigh': a.w_neigh.detach().cpu().double(),
                        'bias': a.bias.detach().cpu().double()}
               for l, per in enumerate(model._aggs) for t, a in per.items()}
    want = omodel.forward(plan, weights, [torch.from_numpy(x) for x in xin], agg=agg)
    for k in range(2):
        assert (out[k].cpu().double() - want[k]).abs().max().item() <= 1e-5, (fan, fu, fi, sizes, agg)
    return eng


@pytest.mark.parametrize('seed', range(14))
def test_forward_engine_random_models_against_the_oracle(cuda, seed):
    """Whatever route the engine picks for the shape (one kernel, padded one kernel,
    fused layers, general kernels; mean or max), embeddings equal the float64 oracle
    applied to the ids the kernels sampled."""
    rng = np.random.default_rng(300 + seed)
    n_hops = int(rng.choice([1, 2, 2, 2, 3]))
    fan = [int(rng.integers(1, [12, 6, 3][h] + 1)) for h in range(n_hops)]
    fu, fi = int(rng.choice([8, 19, 24, 32, 64])), int(rng.choice([5, 18, 32, 64]))
    sizes = [int(rng.choice([4, 16, 32, 64])) for _ in range(n_hops)]
    agg = 'max' if seed % 5 == 4 else 'mean'
    nu, ni, b = int(rng.integers(20, 300)), int(rng.integers(5, 60)), int(rng.integers(1, 40))
    _check_engine_against_oracle(cuda, rng, seed, fan, fu, fi, sizes, agg, nu, ni, b)


@pytest.mark.parametrize('seed', range(10))
def test_one_kernel_forward_over_its_supported_range(cuda, seed):
    """The whole-walk kernel on every width combination it is built for, with fan-outs up to its
    limits (64 first-hop samples, 4096 second-hop samples per walk) and odd batch sizes."""
    rng = np.random.default_rng(500 + seed)
    shapes = [(64, 64, 32, 32), (64, 64, 64, 64), (64, 64, 64, 32), (64, 64, 32, 64), (32, 32, 32, 32), (32, 32, 64, 64),
              (128, 128, 32, 32), (128, 128, 64, 64), (64, 32, 32, 32), (32, 64, 32, 32)]
    fu, fi, h1, h2 = shapes[seed]
    n0 = int(rng.choice([1, 2, 7, 25, 40, 64]))
    n1 = int(rng.choice([1, 3, 10, 33, 64]))
    while n1 > 1 and not kernels.sage2_supported(fu, fi, h1, h2, n0, n1):     # LDS bounds the ids a wave keeps
        n1 //= 2
    assert kernels.sage2_supported(fu, fi, h1, h2, n0, n1), (fu, fi, h1, h2, n0, n1)
    nu, ni, b = int(rng.integers(20, 300)), int(rng.integers(5, 60)), int(rng.choice([1, 3, 17, 64]))
    if n0 * n1 > 1000:
        b = min(b, 5)                                              # keeps the float64 restatement small
    eng = _check_engine_against_oracle(cuda, rng, seed, [n0, n1], fu, fi, [h1, h2], 'mean', nu, ni, b)
    assert eng.walk_kernel, "this shape is expected to take the one-kernel route"


@pytest.mark.parametrize('seed', range(16))
def test_fused_layer_random_shapes_equal_the_general_kernels(cuda, seed):
    """One aggregator layer through the fused kernels (narrow and 128-multiple wide) on random
    widths, segment lists, fan-outs, invalid ids and options, against gather + gather-mean +
    projection (+ l2norm) on the same inputs."""
    rng = np.random.default_rng(400 + seed)
    wide = seed % 2 == 1
    if wide:
        fs, fn = int(rng.choice([64, 96, 128, 256])), int(rng.choice([64, 160, 256]))
        h = int(rng.choice([256, 512]))
        max_child = 32
    else:
        fs, fn, h = [(64, 64, 32), (32, 32, 32), (64, 64, 64), (128, 64, 32), (32, 128, 64)][int(rng.integers(0, 5))]
        max_child = 64
    assert kernels.fused_supported(fs, fn, h, 1), (fs, fn, h)
    n_self, n_neigh = int(rng.integers(1, 4000)), int(rng.integers(1, 4000))
    off_s, off_n = int(rng.integers(0, 50)), int(rng.integers(0, 5000))
    gen = torch.Generator(device=cuda).manual_seed(seed)
    t_self = torch.randn((n_self, fs), device=cuda, generator=gen)
    t_neigh = torch.randn((n_neigh, fn), device=cuda, generator=gen)
    ws = torch.randn((fs, h // 2), device=cuda, generator=gen) * 0.1
    wn = torch.randn((fn, h // 2), device=cuda, generator=gen) * 0.1
    bias = torch.randn(h, device=cuda, generator=gen) * 0.1 if rng.integers(0, 4) else None
    act = str(rng.choice(['relu', 'linear']))
    normalize = bool(rng.integers(0, 2))
    segs, want, want_s, want_m = [], [], [], []
    biggest = 1
    for _ in range(int(rng.integers(1, 5))):
        n_heads = int(rng.choice([1, 2, 127, 128, 129, int(rng.integers(1, 700))]))
        nc = int(rng.integers(1, max_child + 1))
        while nc > 1 and not kernels.fused_supported(fs, fn, h, max(biggest, nc)):   # narrow layers: LDS bounds the fan-out
            nc //= 2
        biggest = max(biggest, nc)
        heads = rng.integers(0, n_self, n_heads).astype(np.int32) + off_s
        kids = rng.integers(0, n_neigh, n_heads * nc).astype(np.int32) + off_n
        kids[rng.random(kids.size) < 0.05] = -1                              # below an isolated node
        kids[rng.random(kids.size) < 0.02] = off_n + n_neigh                 # just past the table
        heads[rng.random(heads.size) < 0.03] = off_s - 1 if off_s else -1    # just before it
        heads, kids = torch.from_numpy(heads).to(cuda), torch.from_numpy(kids).to(cuda)
        segs.append({'head_ids': heads, 'child_ids': kids, 'n_heads': n_heads, 'n_child': nc})
        xs = kernels.gather_rows(t_self, heads, off_s)
        xm = kernels.gather_mean(t_neigh, kids, n_heads, nc, off_n)
        o = kernels.sage_project(xs, xm, ws, wn, bias, act)
        want.append(kernels.l2_normalize(o) if normalize else o)
        want_s.append(xs)
        want_m.append(xm)
    assert kernels.fused_supported(fs, fn, h, biggest), (fs, fn, h, biggest)
    want = torch.cat(want)
    save = bool(rng.integers(0, 2))
    save_s = torch.empty((want.shape[0], fs), device=cuda) if save else None
    save_m = torch.empty((want.shape[0], fn), device=cuda) if save else None
    got = kernels.sage_fused(t_self, off_s, t_neigh, off_n, segs, ws, wn, bias, act, normalize=normalize,
                             save_self=save_s, save_mean=save_m)
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= 2e-5 * scale, (fs, fn, h, act, normalize)
    if save:
        assert torch.equal(save_s, torch.cat(want_s))
        assert float((save_m - torch.cat(want_m)).abs().max()) <= 1e-5 * max(1.0, float(torch.cat(want_m).abs().max()))


@pytest.mark.parametrize('seed', range(12))
def test_three_hop_two_launch_route_random_shapes(cuda, seed):
    """Random fan-outs, widths, batch sizes and graphs (some nodes without neighbors) through the two-launch
    three-hop route (gs_sage2_forward_subtrees + gs_sage3_seeds_forward): ids, error flags and embeddings equal
    those of the sampler + the general kernels."""
    import graphsage_amd as ga
    from graphsage_amd.engine import SageForwardEngine
    rng = np.random.default_rng(900 + seed)
    fu, fi = [(64, 64), (32, 32), (128, 128), (64, 32), (32, 64)][int(rng.integers(0, 5))]
    if (fu, fi) == (64, 64):
        h1, h2 = [(32, 32), (64, 64), (64, 32), (32, 64)][int(rng.integers(0, 4))]
    elif fu == fi:
        h1, h2 = [(32, 32), (64, 64)][int(rng.integers(0, 2))]
    else:
        h1, h2 = 32, 32
    h3 = int(rng.choice([32, 64]))
    fan = [int(rng.integers(1, 31)), int(rng.integers(1, 13)), int(rng.integers(1, 9))]
    nu, ni, b = int(rng.integers(30, 400)), int(rng.integers(10, 90)), int(rng.integers(1, 70))
    ne = int(rng.integers(200, 3000))
    lone_u, lone_i = int(rng.integers(0, 4)), int(rng.integers(0, 3))        # nodes without neighbors
    u = rng.integers(lone_u, nu, ne)
    it = rng.integers(0, ni - lone_i, ne) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, fu)).astype(np.float32),
             'item': rng.standard_normal((ni, fi)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))}, np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, b, fan, head_node_types=['user', 'item'], seed=seed)
    model = ga.GraphSAGE([h1, h2, h3], gen, seed=seed + 1).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.2, 0.2)
    su = rng.integers(0, nu, b)                                    # any user, the lone ones included
    si = rng.integers(0, ni, b) + nu
    seeds = torch.tensor(list(su) + list(si), dtype=torch.int32, device=cuda)
    walk_ids = torch.from_numpy(rng.integers(0, 2 ** 40, 2 * b)).to(cuda)
    eng = SageForwardEngine(g, model, b, keep_ids=True)
    ref = SageForwardEngine(g, model, b, use_fused=False)
    if not eng.subtree_kernel:
        pytest.skip("shape outside the two-launch route: fan-outs %s widths %s" % (fan, (fu, fi, h1, h2, h3)))
    out = [x.clone() for x in eng.forward(seeds, walk_ids, rng_seed=seed + 5)]
    want = [x.clone() for x in ref.forward(seeds, walk_ids, rng_seed=seed + 5)]
    assert eng.err_flag.tolist() == ref.err_flag.tolist()
    for a, c in zip(eng.hops, ref.hops):
        assert torch.equal(a, c), (fan, fu, fi)
    for k in range(2):
        d = (out[k].double() - want[k].double()).abs()
        tol = 1e-5 * want[k].double().abs() + 1e-6 * float(want[k].abs().max())
        assert bool((d <= tol).all()), (fan, fu, fi, h1, h2, h3, float((d / tol).max()))
