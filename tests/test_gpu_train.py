"""Training step on the GPU against the float64 oracle: loss, every gradient,
and the parameters after one Adam update; then a short run must learn."""
import numpy as np
import pytest
import torch

import graphsage_amd as ga
from graphsage_amd.train import Trainer
from oracle import model as omodel

pytestmark = pytest.mark.gpu


def _toy(seed=0, nu=150, ni=40, fu=24, fi=19):
    rng = np.random.default_rng(seed)
    # two user communities and two item communities; positive links inside a community
    cu, ci = rng.integers(0, 2, nu), rng.integers(0, 2, ni)
    u = rng.integers(0, nu, 4000)
    it = rng.integers(0, ni, 4000)
    keep = cu[u] == ci[it]
    u, it = np.concatenate([u[keep], np.arange(nu)]), np.concatenate([it[keep], np.arange(nu) % ni])
    e = np.stack([u, it + nu])
    feats = {'user': (rng.standard_normal((nu, fu)) + cu[:, None]).astype(np.float32),
             'item': (rng.standard_normal((ni, fi)) - ci[:, None]).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))},
                 np.concatenate([e, e[::-1]], axis=1))
    lu, li = rng.integers(0, nu, 600), rng.integers(0, ni, 600)
    links = [(int(a), int(b) + nu) for a, b in zip(lu, li)]
    # additive head (concat -> Dense(1)): a target an additive score can express
    targets = (cu[lu] * ci[li]).astype(np.float32)
    return g, links, targets


def test_one_step_matches_oracle(cuda):
    g, links, targets = _toy()
    b, ns, sizes = 64, [8, 4], [32, 32]
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
    model = ga.GraphSAGE(sizes, gen, seed=11)
    tr = Trainer(model, lr=1e-2)
    seq = gen.flow(links, targets=targets, shuffle=False)
    batch, t = seq[0]
    before = [p.detach().cpu().double().clone() for p in tr.params]

    # oracle: same batch tensors, same weights, float64
    plan = omodel.LayerPlan(sizes, ns, g.node_feature_sizes)
    weights = {(l, ty): {'w_self': a.w_self.detach().cpu().double().requires_grad_(True),
                         'w_neigh': a.w_neigh.detach().cpu().double().requires_grad_(True),
                         'bias': a.bias.detach().cpu().double().requires_grad_(True)}
               for l, aggs in enumerate(model._aggs) for ty, a in aggs.items()}
    kernel = tr.head.kernel.detach().cpu().double().requires_grad_(True)
    hbias = tr.head.bias.detach().cpu().double().requires_grad_(True)
    emb = omodel.forward(plan, weights, [torch.from_numpy(x).double() for x in batch.numpy()])
    want_loss = omodel.mse_loss(omodel.link_head(emb[0], emb[1], kernel, hbias), torch.from_numpy(t).double())
    want_loss.backward()

    loss, pred = tr.step(batch, t)
    want_loss = float(want_loss.detach())
    assert abs(float(loss.detach()) - want_loss) < 1e-5 * max(1.0, want_loss)
    ref_params = [weights[(l, ty)][n] for l, aggs in enumerate(model._aggs) for ty, a in aggs.items()
                  for n in ('w_self', 'w_neigh', 'bias')] + [kernel, hbias]
    assert len(ref_params) == len(tr.params)
    off = 0
    for p, ref, old in zip(tr.params, ref_params, before):
        grad = tr.bucket.flat[off:off + p.numel()].view_as(p).cpu().double()
        off += p.numel()
        scale = float(ref.grad.abs().max()) + 1e-12
        assert float((grad - ref.grad).abs().max()) <= 2e-5 * scale
        # first Adam step: p - lr * g / (|g| + eps')  with bias correction
        m, v = 0.1 * ref.grad, 0.001 * ref.grad ** 2
        want = old - 1e-2 * (1 - 0.999) ** 0.5 / (1 - 0.9) * m / (v.sqrt() + 1e-7)
        got = p.detach().cpu().double()
        big = ref.grad.abs() > 1e-4 * scale                    # sign of tiny gradients is rounding noise
        assert float((got - want)[big].abs().max()) < 1e-4 if big.any() else True


@pytest.mark.parametrize('rate', [0.0, 0.2])
def test_short_run_learns(cuda, rate):
    """(rate > 0: the plan's own dropout, through fit-style steps on padded 24/19-wide tables)"""
    g, links, targets = _toy(1)
    gen = ga.GraphSAGEGenerator(g, 100, [8, 4], head_node_types=['user', 'item'], seed=5)
    model = ga.GraphSAGE([32, 32], gen, dropout=rate, seed=2)
    tr = Trainer(model, lr=1e-2) if not rate else Trainer(model, lr=1e-2, graph=g, batch_size=100)
    assert (tr.engine is not None) == bool(rate)
    seq = gen.flow(links, targets=targets, shuffle=True, seed=1)
    losses = []
    if rate:
        losses = tr.fit(seq, epochs=15, verbose=0)['loss']       # full batches: plan steps with dropout
        assert tr._n_plan_steps > 0
    for epoch in range(0 if rate else 15):
        tot = 0.0
        for batch, t in seq:
            loss, _ = tr.step(batch, t)
            tot += float(loss)
        losses.append(tot / len(seq))
        seq.on_epoch_end()
    assert losses[-1] < 0.7 * losses[0], losses
    preds, ys = [], []
    model.eval()
    with torch.no_grad():
        for batch, t in gen.flow(links, targets=targets, shuffle=False):
            preds.append(tr.head(model(batch)).cpu().numpy())
            ys.append(t)
    auc = omodel.keras_auc(np.concatenate(ys), np.concatenate(preds))
    assert auc > 0.8, auc


@pytest.mark.parametrize('ns,sizes,f', [([8, 4], [32, 32], 64), ([5, 3, 2], [32, 64, 32], 32), ([25, 10], [32, 32], 64),
                                        ([3, 2], [32, 32], 128), ([10, 5], [64, 32], 32), ([7], [32], 64),
                                        ([2, 2, 2, 2], [32, 32, 32, 32], 32), ([12, 1], [32, 64], 64),
                                        ([8, 4], [32, 32], (24, 19)), ([5, 3, 2], [32, 64, 32], (28, 18)),
                                        ([5, 3, 2], [256, 256, 256], 64), ([4, 3], [256, 256], (128, 96)),   # wide layers
                                        # two hops over the other widths of the one-kernel forward; unequal feature
                                        # widths take the backward route without paired weight gradients
                                        ([6, 3], [64, 64], 64), ([6, 3], [64, 64], 32), ([4, 2], [64, 64], 128),
                                        ([5, 2], [32, 32], (64, 32)), ([5, 2], [32, 32], (32, 64)),
                                        ([5, 2], [64, 32], 64), ([5, 2], [32, 64], 64),
                                        # three hops of 32 units: the two-kernel backward pass on the layered tables
                                        ([5, 3, 2], [32, 32, 32], 64), ([4, 6, 3], [32, 32, 32], 32), ([2, 17, 2], [32, 32, 32], 64)])
def test_plan_engine_step_equals_autograd_step(cuda, ns, sizes, f):
    """The static training plan (fused forward with saved activations + kernel
    backward) produces the same loss, gradients and updated weights as the
    autograd path on the same sampled batch."""
    # (odd widths, the last two cases: the plan runs on zero-padded tables and layer-0 weights)
    fu, fi = f if isinstance(f, tuple) else (f, f)
    g, links, targets = _toy(3, fu=fu, fi=fi)
    b = 48
    results = []
    for engine in ('autograd', 'plan'):
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE(sizes, gen, seed=11)
        from graphsage_amd.train import LinkHead
        model.build()
        head = LinkHead(sizes[-1], cuda, seed=5)
        tr = Trainer(model, head=head, lr=1e-2, engine=engine, graph=g, batch_size=b)
        lk = np.asarray(links[:b])
        seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
        walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda)
        t = torch.from_numpy(targets[:b]).to(cuda)
        if engine == 'plan':
            assert tr.engine is not None
            loss, _ = tr.step_device(seeds, t, walk_ids, rng_seed=77)
        else:
            # same draws through the generator-level API
            hops, _ = __import__('graphsage_amd').kernels.sample(
                g.device_tensors()['indptr'], g.device_tensors()['indices'], seeds, ns, rng_seed=77, walk_ids=walk_ids)
            slot_ids = [seeds[:b].reshape(b, 1), seeds[b:].reshape(b, 1)]
            for hop in hops:
                slot_ids += [hop[:b], hop[b:]]
            batch = ga.BatchFeatures(g, b, slot_ids, gen._slot_types)
            loss, _ = tr.step(batch, targets[:b])
        results.append((float(loss), tr.bucket.flat.clone(), [p.detach().clone() for p in tr.params]))
    (l0, g0, p0), (l1, g1, p1) = results
    assert abs(l0 - l1) < 1e-6 * max(1.0, abs(l0))
    scale = float(g0.abs().max())
    assert float((g0 - g1).abs().max()) <= 2e-5 * scale
    for a, c in zip(p0, p1):
        assert float((a - c).abs().max()) < 2e-4
    if isinstance(f, tuple):                        # the copies in and out of the padded weights replay from a graph too
        before = [p.detach().clone() for p in tr.params]
        loss_g, _ = tr.step_graph(seeds, t, walk_ids, rng_seed=77)
        assert np.isfinite(float(loss_g)) and float(loss_g) < l1 + 1e-3
        assert any(float((a - c.detach()).abs().max()) > 0 for a, c in zip(before, tr.params))


def test_link_head_kernel_and_adam_kernel(cuda):
    from graphsage_amd import kernels
    gen = torch.Generator().manual_seed(0)
    for b, e in [(100, 32), (4096, 32), (7, 64), (33, 100)]:
        u = torch.randn(b, e, generator=gen, dtype=torch.float64, requires_grad=True)
        i = torch.randn(b, e, generator=gen, dtype=torch.float64, requires_grad=True)
        k = (torch.randn(2 * e, 1, generator=gen, dtype=torch.float64) * 0.3).requires_grad_(True)
        bias = torch.tensor([0.1], dtype=torch.float64, requires_grad=True)
        y = (torch.rand(b, generator=gen) > 0.5).double()
        pred = omodel.link_head(u, i, k, bias)
        loss = omodel.mse_loss(pred, y)
        loss.backward()
        f = lambda t: t.detach().float().to(cuda).contiguous()
        gk, gb = torch.zeros(2 * e, 1, device=cuda), torch.zeros(1, device=cuda)
        step = kernels.LinkHeadStep(b, e, f(k), f(bias), gk, gb, cuda)
        l, p = step.run(f(u), f(i), f(y))
        assert abs(float(l) - float(loss)) < 1e-6
        assert float((p.cpu().double() - pred.detach().reshape(-1)).abs().max()) < 1e-6
        for got, want in ((step.g_user, u.grad), (step.g_item, i.grad), (gk, k.grad), (gb, bias.grad)):
            sc = float(want.abs().max())
            assert float((got.cpu().double().reshape(want.shape) - want).abs().max()) <= 1e-5 * sc
        assert torch.equal(kernels.link_head(f(u), f(i), f(k), f(bias)), p)
    # Adam kernel against the update rule
    n = 1000
    p0 = torch.randn(n, generator=gen)
    g = torch.randn(n, generator=gen)
    pf, m, v = p0.clone().to(cuda), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    x, mm, vv = p0.double().numpy().copy(), np.zeros(n), np.zeros(n)
    for t in range(1, 4):
        lr_t = 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        kernels.adam_step(pf, g.to(cuda), m, v, lr_t, 0.9, 0.999, 1e-7)
        mm = 0.9 * mm + 0.1 * g.double().numpy()
        vv = 0.999 * vv + 0.001 * g.double().numpy() ** 2
        x = x - lr_t * mm / (np.sqrt(vv) + 1e-7)
        np.testing.assert_allclose(pf.cpu().numpy(), x, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize('rate,ns,sizes', [(0.0, [6, 3], [32, 32]), (0.2, [6, 3], [32, 32]), (0.0, [4, 3, 2], [32, 32, 32])])
def test_graph_replayed_steps_equal_eager_steps(cuda, rate, ns, sizes):
    """(with dropout as well: the masks are a function of key, walk and position, so a replayed step draws what
    the eager step draws; three hops: the two-launch forward and the two-kernel backward pass)"""
    g, links, targets = _toy(4, fu=64, fi=64)
    b = 32
    lk = np.asarray(links)
    outs = []
    for mode in ('eager', 'graph'):
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE(sizes, gen, dropout=rate, seed=11).build()
        from graphsage_amd.train import LinkHead
        tr = Trainer(model, head=LinkHead(32, cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=b)
        assert len(ns) == 2 or tr.engine.subtree_train
        losses = []
        for k in range(5):
            part = lk[k * b:(k + 1) * b]
            seeds = torch.from_numpy(np.concatenate([part[:, 0], part[:, 1]]).astype(np.int32)).to(cuda)
            wid = torch.arange(2 * b, dtype=torch.int64, device=cuda) + 1000 * k
            t = torch.from_numpy(targets[k * b:(k + 1) * b]).to(cuda)
            fn = tr.step_device if mode == 'eager' else tr.step_graph
            loss, _ = fn(seeds, t, wid, rng_seed=9)
            losses.append(float(loss))
        outs.append((losses, [p.detach().clone() for p in tr.params]))
    (l0, p0), (l1, p1) = outs
    assert l0 == l1, (l0, l1)                                # same kernels, same order: same bits
    for a, c in zip(p0, p1):
        assert torch.equal(a, c)


# ------------------------------------------- MovieLens-format run end to end ----

def _movielens_files(tmp_path, n_users=300, n_movies=120, seed=0):
    rng = np.random.default_rng(seed)
    genres = ['Action', 'Comedy', 'Drama', 'Horror', 'Romance', 'Sci-Fi', "Children's"]
    ages = [1, 18, 25, 35, 45, 50, 56]
    with open(tmp_path / 'users.dat', 'w') as fh:
        for u in range(1, n_users + 1):
            fh.write("%d::%s::%d::%d::%05d\n" % (u, 'FM'[rng.integers(2)], ages[rng.integers(7)], rng.integers(21),
                                                 rng.integers(99999)))
    with open(tmp_path / 'movies.dat', 'w', encoding='iso-8859-1') as fh:
        for m in range(1, n_movies + 1):
            k = rng.integers(1, 4)
            fh.write("%d::Film \xe9 %d (1999)::%s\n" % (m, m, '|'.join(rng.choice(genres, k, replace=False))))
    return str(tmp_path / 'users.dat'), str(tmp_path / 'movies.dat')


def test_movielens_format_fit_learns_and_both_step_paths_agree(cuda, tmp_path):
    from graphsage_amd import metrics, movielens
    from graphsage_amd.link_utils import link_classification
    users, movies = _movielens_files(tmp_path)
    ratings = movielens.synthetic_ratings(users, movies, 12_000, seed=1)
    data = movielens.load_movielens(users, movies, ratings, pad_to=32)
    g = ga.Graph(node_features=data['node_features'], nodes=data['nodes'], edges=data['edges'])
    links = [tuple(l) for l in data['links'].tolist()]
    tr_idx, te_idx = movielens.train_test_split(len(links), 0.7, 0.3, seed=0)

    def run(engine, epochs):
        gen = ga.GraphSAGEGenerator(g, 100, [8, 4], head_node_types=['user', 'item'], seed=3)
        train = gen.flow([links[i] for i in tr_idx], targets=data['labels'][tr_idx], shuffle=True, seed=4)
        test = gen.flow([links[i] for i in te_idx], targets=data['labels'][te_idx])
        model = ga.GraphSAGE([32, 32], gen, seed=7)
        _, x_out = model.in_out_tensors()
        head = link_classification(edge_embedding_method='concat')(x_out, seed=8)
        t = Trainer(model, head=head, lr=1e-2, graph=g, batch_size=100, engine=engine)
        return t, t.fit(train, validation_data=test, epochs=epochs, verbose=0), test

    t_plan, h_plan, test = run('plan', 4)
    assert t_plan.engine is not None
    assert h_plan['loss'][-1] < h_plan['loss'][0] and h_plan['val_auc'][-1] > 0.6, h_plan
    assert set(h_plan) >= {'loss', 'auc', 'binary_accuracy', 'root_mean_squared_error', 'val_loss', 'val_auc'}
    # same run through autograd: the two step paths are the same computation
    t_auto, h_auto, _ = run('autograd', 1)
    assert t_auto.engine is None
    assert abs(h_auto['loss'][0] - h_plan['loss'][0]) <= 1e-4 * h_plan['loss'][0]
    # the streaming AUC equals the oracle's restatement of the Keras metric on the same scores
    pred = t_plan.predict(test).cpu().numpy()
    y = np.concatenate([test[i][1] for i in range(len(test))])
    m = metrics.AUC()
    m.update_state(y[:500], torch.from_numpy(pred[:500]).to(cuda))
    m.update_state(y[500:], torch.from_numpy(pred[500:]).to(cuda))
    assert abs(m.result() - omodel.keras_auc(y, pred)) < 1e-6
    ev = t_plan.evaluate(test)
    assert abs(ev['auc'] - h_plan['val_auc'][-1]) < 1e-9
    with pytest.raises(NotImplementedError):
        link_classification(edge_embedding_method='no-such-operator')
    # another operator of link_utils.py: a torch head, trained through autograd
    gen = ga.GraphSAGEGenerator(g, 100, [8, 4], head_node_types=['user', 'item'], seed=3)
    train = gen.flow([links[i] for i in tr_idx[:2000]], targets=data['labels'][tr_idx[:2000]], shuffle=True, seed=4)
    model = ga.GraphSAGE([32, 32], gen, seed=7)
    head = link_classification(edge_embedding_method='mul')(model.in_out_tensors()[1], seed=8)
    t_mul = Trainer(model, head=head, lr=1e-2, graph=g, batch_size=100)
    assert t_mul.engine is None
    h_mul = t_mul.fit(train, epochs=3, verbose=0)
    assert h_mul['loss'][-1] < h_mul['loss'][0]


def test_fit_surfaces_a_node_without_neighbors_like_the_reference(cuda):
    """Node 2 (a user) has no edge: the reference raises while it builds the batch
    (sampler.py:81-82 / bipartite_graph.py:63); the planned path flags it in the
    kernel and fit raises after the pass."""
    rng = np.random.default_rng(0)
    nu, ni = 40, 10
    u = np.concatenate([np.arange(nu), rng.integers(0, nu, 300)])
    u = u[u != 2]
    it = rng.integers(0, ni, u.size) + nu
    e = np.stack([u, it])
    feats = {'user': rng.standard_normal((nu, 32)).astype(np.float32),
             'item': rng.standard_normal((ni, 32)).astype(np.float32)}
    g = ga.Graph(feats, {'user': list(range(nu)), 'item': list(range(nu, nu + ni))}, np.concatenate([e, e[::-1]], axis=1))
    gen = ga.GraphSAGEGenerator(g, 8, [3, 2], head_node_types=['user', 'item'], seed=1)
    links = [(int(a), int(b)) for a, b in zip(u[:15], it[:15])] + [(2, nu)]
    seq = gen.flow(links, targets=np.ones(16, np.float32))
    model = ga.GraphSAGE([32, 32], gen, seed=2)
    tr = Trainer(model, lr=1e-2, graph=g, batch_size=8)
    assert tr.engine is not None
    with pytest.raises(ValueError) as ei:
        tr.fit(seq, epochs=1, verbose=0)
    assert str(ei.value).startswith("Samples must exist. Self-node also can be a sample")
    with pytest.raises(KeyError):                      # the same error is a KeyError too, as the dict lookup gives
        tr.evaluate(seq)


def test_staged_epoch_rows_equal_the_per_batch_roots(cuda):
    graph, links, y = _toy(seed=5)
    gen = ga.GraphSAGEGenerator(graph, 16, [3, 2], head_node_types=['user', 'item'], seed=9)
    seq = gen.flow(links, targets=y, shuffle=True, seed=2)
    for _ in range(2):                                           # a second epoch re-stages in the new order
        st = seq.epoch_on_device()
        assert st['n_batches'] == len(links) // 16 and st is seq.epoch_on_device()
        for i in range(st['n_batches']):
            seeds, walk_ids, key, t = seq.roots(i)
            assert torch.equal(st['roots'][i], seeds) and torch.equal(st['walk_ids'][i], walk_ids)
            assert key == st['key']
            np.testing.assert_array_equal(st['targets'][i].cpu().numpy(), np.asarray(t, dtype=np.float32))
        seq.on_epoch_end()
    bad = gen.flow(links[:31] + [(links[0][0], 10 ** 6)], targets=np.zeros(32), shuffle=False)
    with pytest.raises(KeyError):                                # unknown node in a later link, as roots() raises
        bad.epoch_on_device()


def test_dropout_models_and_the_training_plan(cuda):
    """Dropout on every aggregator input (models.py:152,165) is drawn inside the two-hop one-kernel route of the
    static plan; other depths have no such kernel: 'auto' takes autograd for them, 'plan' refuses."""
    g, links, targets = _toy(1, fu=32, fi=32)
    gen2 = ga.GraphSAGEGenerator(g, 16, [3, 2], head_node_types=['user', 'item'], seed=1)
    gen3 = ga.GraphSAGEGenerator(g, 16, [3, 2, 2], head_node_types=['user', 'item'], seed=1)
    tr = Trainer(ga.GraphSAGE([32, 32], gen2, dropout=0.5, seed=2), lr=1e-2, graph=g, batch_size=16)
    assert tr.engine is not None and tr.engine.dropout == 0.5 and tr.engine.self_ids is None
    deep = ga.GraphSAGE([32, 32, 32], gen3, dropout=0.5, seed=2)
    assert Trainer(deep, lr=1e-2, graph=g, batch_size=16).engine is None
    with pytest.raises(Exception):
        Trainer(deep, lr=1e-2, engine='plan', graph=g, batch_size=16)
    with pytest.raises(Exception):
        Trainer(ga.GraphSAGE([32, 32], gen2, dropout=1.0, seed=2), lr=1e-2, engine='plan', graph=g, batch_size=16)
    assert Trainer(ga.GraphSAGE([32, 32, 32], gen3, seed=2), lr=1e-2, graph=g, batch_size=16).engine is not None


@pytest.mark.parametrize('ns,sizes,f,rate', [([6, 3], [32, 32], 64, 0.3), ([25, 10], [32, 32], 64, 0.1), ([5, 4], [64, 64], 32, 0.5),
                                             ([3, 2], [32, 32], 128, 0.25)])
def test_plan_step_with_dropout_equals_the_oracle_fed_the_same_masks(cuda, ns, sizes, f, rate):
    """One optimisation step of the static plan with dropout > 0 against the float64 restatement given the masks
    the device's rule defines (oracle/dropout.py): loss, predictions, every gradient.  A second step on the same
    inputs repeats the first bit for bit (the masks are a function of seed, walk and position), another seed
    does not, and rate 0 leaves the step what it was."""
    from graphsage_amd.train import LinkHead
    from graphsage_amd import kernels
    from oracle import dropout as odrop
    g, links, targets = _toy(3, fu=f, fi=f)
    b = 40
    lk = np.asarray(links[:b])
    seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
    walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda) * 3 + (1 << 34)
    t = torch.from_numpy(targets[:b]).to(cuda)
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)

    def step(drop, rng_seed, lr=0.0):
        model = ga.GraphSAGE(sizes, gen, dropout=drop, seed=11).build()
        bias_gen = torch.Generator().manual_seed(1)
        for p in model.parameters():
            if p.dim() == 1:                                 # a zero bias would hide a misplaced one
                p.data.copy_(torch.rand(p.shape, generator=bias_gen) * 0.2 - 0.1)
        tr = Trainer(model, head=LinkHead(sizes[-1], cuda, seed=5), lr=lr, engine='plan', graph=g, batch_size=b)
        loss, pred = tr.step_device(seeds, t, walk_ids, rng_seed=rng_seed)
        assert tr.engine.err_flag.tolist() == [0, 0]
        return tr, model, float(loss), pred.detach().reshape(-1).cpu().clone(), tr.bucket.flat.detach().cpu().clone()

    tr, model, loss, pred, grads = step(rate, 77)
    # the oracle: the same draws (gs_sample), rows by torch indexing, float64, masks by the stated rule
    dev = g.device_tensors()
    hops, _ = kernels.sample(dev['indptr'], dev['indices'], seeds, ns, rng_seed=77, walk_ids=walk_ids)
    slot_ids = [seeds[:b].reshape(b, 1), seeds[b:].reshape(b, 1)]
    for hop in hops:
        slot_ids += [hop[:b], hop[b:]]
    nu = g.num_users
    xin = []
    for s_, ids in enumerate(slot_ids):
        ty = ('user', 'item')[(s_ % 2 + s_ // 2) % 2]
        x = dev[ty][ids.reshape(-1).long() - (nu if ty == 'item' else 0)].cpu().double()
        xin.append(x.reshape(b, -1, x.shape[1]))
    plan = omodel.LayerPlan(sizes, ns, g.node_feature_sizes)
    weights = {(l, ty): {'w_self': a.w_self.detach().cpu().double().requires_grad_(True),
                         'w_neigh': a.w_neigh.detach().cpu().double().requires_grad_(True),
                         'bias': a.bias.detach().cpu().double().requires_grad_(True)}
               for l, aggs in enumerate(model._aggs) for ty, a in aggs.items()}
    kernel = tr.head.kernel.detach().cpu().double().requires_grad_(True)
    hbias = tr.head.bias.detach().cpu().double().requires_grad_(True)
    hook = odrop.two_hop(walk_ids.cpu().numpy(), b, ns[0], ns[1], rate, kernels.dropout_seed_of(77))
    emb = omodel.forward(plan, weights, xin, dropout=hook)
    want_pred = omodel.link_head(emb[0], emb[1], kernel, hbias)
    want_loss = omodel.mse_loss(want_pred, t.cpu().double())
    want_loss.backward()
    assert abs(loss - float(want_loss)) < 1e-5 * max(1.0, float(want_loss))
    assert float((pred.double() - want_pred.detach().reshape(-1)).abs().max()) < 2e-6
    ref = [weights[(l, ty)][n] for l, aggs in enumerate(model._aggs) for ty, a in aggs.items()
           for n in ('w_self', 'w_neigh', 'bias')] + [kernel, hbias]
    want_grads = torch.cat([r.grad.reshape(-1) for r in ref])
    assert want_grads.numel() == grads.numel()
    off = 0
    for r in ref:                                        # per parameter: its own scale
        gr, wg = grads[off:off + r.numel()].double(), r.grad.reshape(-1)
        off += r.numel()
        assert float((gr - wg).abs().max()) <= 2e-5 * (float(wg.abs().max()) + 1e-12)
    # the masks really drop: without them the prediction differs
    emb0 = omodel.forward(plan, weights, xin)
    assert float((omodel.link_head(emb0[0], emb0[1], kernel, hbias).detach().reshape(-1) - pred.double()).abs().max()) > 1e-3
    # deterministic in (seed, walk, position); another seed draws other masks
    _, _, loss2, pred2, grads2 = step(rate, 77)
    assert loss2 == loss and torch.equal(pred2, pred) and torch.equal(grads2, grads)
    _, _, _, pred3, _ = step(rate, 78)
    assert not torch.equal(pred3, pred)
    # rate 0 is the plan without dropout (ids instead of rows for the own rows, same numbers as before)
    tr0, _, loss0, pred0, grads0 = step(0.0, 77)
    assert tr0.engine.self_ids is not None or f != 64 or sizes[0] != 32
    assert float((pred0.double() - omodel.link_head(emb0[0], emb0[1], kernel, hbias).detach().reshape(-1)).abs().max()) < 2e-6


def test_plan_step_without_bias_and_normalisation_equals_autograd(cuda):
    g, links, targets = _toy(4, fu=64, fi=64)
    b, ns = 32, [6, 3]
    results = []
    for engine in ('autograd', 'plan'):
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE([32, 32], gen, bias=False, normalize=None, seed=11).build()
        from graphsage_amd.train import LinkHead
        tr = Trainer(model, head=LinkHead(32, cuda, seed=5), lr=1e-2, engine=engine, graph=g, batch_size=b)
        lk = np.asarray(links[:b])
        seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
        walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda)
        t = torch.from_numpy(targets[:b]).to(cuda)
        if engine == 'plan':
            assert tr.engine is not None
            loss, _ = tr.step_device(seeds, t, walk_ids, rng_seed=9)
        else:
            hops, _ = ga.kernels.sample(g.device_tensors()['indptr'], g.device_tensors()['indices'], seeds, ns,
                                        rng_seed=9, walk_ids=walk_ids)
            slot_ids = [seeds[:b].reshape(b, 1), seeds[b:].reshape(b, 1)]
            for hop in hops:
                slot_ids += [hop[:b], hop[b:]]
            loss, _ = tr.step(ga.BatchFeatures(g, b, slot_ids, gen._slot_types), targets[:b])
        results.append((float(loss), tr.bucket.flat.clone()))
    (l0, g0), (l1, g1) = results
    assert abs(l0 - l1) < 1e-6 * max(1.0, abs(l0))
    assert float((g0 - g1).abs().max()) <= 2e-5 * float(g0.abs().max())


@pytest.mark.parametrize('b,n0,h1,h2,normalize,act', [(48, 8, 32, 32, True, 'linear'), (5, 25, 64, 64, True, 'linear'),
                                                      (130, 3, 32, 64, False, 'relu'), (1, 1, 64, 32, True, 'relu'),
                                                      (4096, 25, 32, 32, True, 'linear')])
def test_backward_tail_kernel_equals_the_layer_by_layer_kernels(cuda, b, n0, h1, h2, normalize, act):
    """gs_sage2_backward_tail against gs_l2_normalize_bwd + the input gradients of
    gs_sage_project_bwd + gs_gather_mean_bwd on the same inputs, both node types."""
    from graphsage_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(b + n0)
    rnd = lambda *shape: torch.randn(shape, device=cuda, generator=gen)
    types = ('user', 'item')
    other = {'user': 'item', 'item': 'user'}
    z = {t: rnd(b, h2) for t in types}
    if normalize:
        z['user'][0] = 0.0                                   # a zero row: the normalisation's guard
    g_emb = {t: rnd(b, h2) for t in types}
    ws = {t: rnd(h1, h2 // 2) * 0.2 for t in types}
    wn = {t: rnd(h1, h2 // 2) * 0.2 for t in types}
    g_z = {t: torch.empty(b, h2, device=cuda) for t in types}
    g_h1 = {t: torch.full((b * (1 + n0), h1), float('nan'), device=cuda) for t in types}
    kernels.sage2_backward_tail(z, g_emb, ws, wn, n0, normalize, act, g_z, g_h1)
    for t in types:
        want_gz = kernels.l2_normalize_bwd(z[t], g_emb[t]) if normalize else g_emb[t]
        row_scale = want_gz.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)     # (fused multiply-adds round differently)
        assert float(((g_z[t] - want_gz).abs() / row_scale).max()) <= 2e-6
        x = torch.zeros(b, h1, device=cuda)                  # the weight-gradient inputs do not matter here
        _, _, _, gxs, gxm = kernels.sage_project_bwd(x, x, ws[t], wn[t], act, z[t], g_z[t], need_bias=False)
        scale = max(1.0, float(gxs.abs().max()), float(gxm.abs().max()))
        assert float((g_h1[t][:b] - gxs).abs().max()) <= 2e-6 * scale
        spread = torch.empty(b * n0, h1, device=cuda)
        kernels.gather_mean_bwd(gxm, None, n0, spread)
        assert float((g_h1[other[t]][b:] - spread).abs().max()) <= 2e-6 * scale
    assert not any(bool(torch.isnan(g_h1[t]).any()) for t in types)   # every row written


def _pack_mask(h):
    """[rows, H] > 0 packed the way gs_sage2_forward packs it: bit c % 32 of word row * H/32 + c / 32."""
    bits = (h > 0).to(torch.int64).reshape(h.shape[0], h.shape[1] // 32, 32)
    words = (bits << torch.arange(32, device=h.device)).sum(-1)
    return (words & 0xFFFFFFFF).to(torch.int64)


@pytest.mark.parametrize('b,n0,n1,f,h1,h2', [(24, 8, 4, 64, 32, 32), (9, 25, 10, 64, 64, 32), (130, 3, 2, 64, 32, 64)])
def test_saving_forward_writes_relu_mask_words_and_the_tail_kernel_applies_them(cuda, b, n0, n1, f, h1, h2):
    """gs_sage2_saved.h1_mask: the hop-1 heads' layer-0 outputs leave as one bit per element (and their
    rows are not written); everything else the forward saves is unchanged.  gs_sage2_backward_tail with
    l0_mask = its plain output times the layer-0 relu masks."""
    from graphsage_amd import kernels
    g, links, _ = _toy(3, fu=f, fi=f)
    dev_t = g.device_tensors()
    gen = torch.Generator(device=cuda).manual_seed(b + n0)
    rnd = lambda *shape: torch.randn(shape, device=cuda, generator=gen)
    types = ('user', 'item')
    weights = {}
    for t in types:
        weights[(0, t)] = (rnd(f, h1 // 2) * 0.2, rnd(f, h1 // 2) * 0.2, rnd(h1) * 0.1)
        weights[(1, t)] = (rnd(h1, h2 // 2) * 0.2, rnd(h1, h2 // 2) * 0.2, rnd(h2) * 0.1)
    lk = np.asarray(links[:b])
    seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
    walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda)
    err = torch.zeros(2, dtype=torch.int32, device=cuda)
    rows = b * (1 + n0)

    def tables():
        nan = lambda *shape: torch.full(shape, float('nan'), device=cuda)
        return {'x_self': {t: nan(rows, f) for t in types}, 'x_mean': {t: nan(rows, f) for t in types},
                'h1': {t: nan(rows, h1) for t in types}, 'x_mean1': {t: nan(b, h1) for t in types},
                'z': {t: nan(b, h2) for t in types}}
    full, lean = tables(), tables()
    lean['h1_mask'] = {t: torch.full((b * n0 * (h1 // 32),), -1, dtype=torch.int32, device=cuda) for t in types}
    outs = []
    for saved in (full, lean):
        call = kernels.Sage2Call(dev_t['indptr'], dev_t['indices'], dev_t['user'], dev_t['item'], [n0, n1], weights,
                                 True, b, saved=saved)
        outs.append(call.run(seeds, err, rng_seed=4, walk_ids=walk_ids).clone())
    assert err.tolist() == [0, 0] and torch.equal(outs[0], outs[1])
    for t in types:
        for name in ('x_self', 'x_mean', 'x_mean1', 'z'):
            assert torch.equal(full[name][t], lean[name][t])
        assert torch.equal(full['h1'][t][:b], lean['h1'][t][:b])                 # the roots' rows: still values
        assert bool(torch.isnan(lean['h1'][t][b:]).all())                        # the hop-1 heads' rows: not written
        got = lean['h1_mask'][t].to(torch.int64) & 0xFFFFFFFF
        assert torch.equal(got, _pack_mask(full['h1'][t][b:]).reshape(-1))
        assert 0.05 < float((full['h1'][t][b:] > 0).float().mean()) < 0.95       # (masks that mean something)

    z = full['z']
    g_emb = {t: rnd(b, h2) for t in types}
    ws, wn = {t: weights[(1, t)][0] for t in types}, {t: weights[(1, t)][1] for t in types}
    plain = {t: torch.empty(rows, h1, device=cuda) for t in types}
    masked = {t: torch.full((rows, h1), float('nan'), device=cuda) for t in types}
    gz0, gz1 = ({t: torch.empty(b, h2, device=cuda) for t in types} for _ in range(2))
    kernels.sage2_backward_tail(z, g_emb, ws, wn, n0, True, 'linear', gz0, plain)
    kernels.sage2_backward_tail(z, g_emb, ws, wn, n0, True, 'linear', gz1, masked,
                                l0_mask={'h1': lean['h1'], 'h1_mask': lean['h1_mask']})
    for t in types:
        assert torch.equal(gz0[t], gz1[t])
        assert torch.equal(masked[t], plain[t] * (full['h1'][t] > 0))


def test_plan_backward_with_and_without_the_tail_kernel(cuda):
    """A two-hop training step through the one-launch top of the backward pass and through the
    layer-by-layer kernels: same loss, same gradients."""
    from graphsage_amd.train import LinkHead
    g, links, targets = _toy(5, fu=64, fi=64)
    b, ns, sizes = 40, [6, 3], [32, 32]
    grads = []
    for tail in (True, False):
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE(sizes, gen, seed=11)
        model.build()
        tr = Trainer(model, head=LinkHead(sizes[-1], cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=b)
        assert tr.engine.tail_kernel
        tr.engine.tail_kernel = tail
        lk = np.asarray(links[:b])
        seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
        loss, _ = tr.step_device(seeds, torch.from_numpy(targets[:b]).to(cuda),
                                 torch.arange(2 * b, dtype=torch.int64, device=cuda), rng_seed=9)
        grads.append((float(loss), tr.bucket.flat.clone()))
    # (with the tail kernel the head's loss is summed per block of walks, without it per block of rows)
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * max(1.0, abs(grads[1][0]))
    scale = float(grads[1][1].abs().max())
    assert float((grads[0][1] - grads[1][1]).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize('b,ns,f,rate,bias,normalize', [(40, [6, 3], 64, 0.0, True, True), (130, [25, 10], 64, 0.0, True, True),
                                                        (7, [1, 2], 32, 0.0, True, True), (48, [4, 2], 32, 0.0, False, False),
                                                        (33, [9, 2], 64, 0.25, True, True), (2100, [5, 2], 64, 0.0, True, True)])
def test_whole_backward_in_one_kernel_equals_the_separate_launches(cuda, b, ns, f, rate, bias, normalize):
    """A two-hop training step with the whole backward pass in one kernel (head, layer 1, every weight
    gradient; one reduction after it) and with the separate launches (top of the backward pass, paired weight
    gradients per layer, their reductions): same loss, same gradients up to the order of the sums, and the
    one-kernel form gives the same bits when it runs again."""
    from graphsage_amd.train import LinkHead
    g, links, targets = _toy(5, fu=f, fi=f)
    pick = np.arange(b) % len(links)                       # (more pairs than links: links repeat, walks still differ)
    sizes = [32, 32]
    grads = []
    for fused in (True, False, True):
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE(sizes, gen, seed=11, bias=bias, normalize='l2' if normalize else None, dropout=rate)
        model.build()
        tr = Trainer(model, head=LinkHead(sizes[-1], cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=b)
        assert tr.engine.tail_kernel and tr.engine.fused_backward
        tr.engine.fused_backward = fused
        lk = np.asarray(links)[pick]
        seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
        loss, pred = tr.step_device(seeds, torch.from_numpy(targets[pick]).to(cuda),
                                    torch.arange(2 * b, dtype=torch.int64, device=cuda), rng_seed=9)
        grads.append((float(loss), tr.bucket.flat.clone(), pred.clone()))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * max(1.0, abs(grads[1][0]))
    assert float((grads[0][2] - grads[1][2]).abs().max()) < 1e-6     # (the same arithmetic, contracted differently)
    scale = float(grads[1][1].abs().max())
    assert scale > 0
    assert float((grads[0][1] - grads[1][1]).abs().max()) <= 2e-6 * scale
    assert grads[0][0] == grads[2][0] and torch.equal(grads[0][1], grads[2][1])


@pytest.mark.parametrize('mode', ['device', 'graph'])
def test_optimizer_step_inside_the_backward_reduction_equals_the_separate_launch(cuda, mode):
    """Three steps with Adam applied by the reduction of the one-kernel backward pass and three with the
    update as a launch of its own: the same weights and the same moments."""
    from graphsage_amd.train import LinkHead
    g, links, targets = _toy(5, fu=64, fi=64)
    b, ns, sizes = 40, [6, 3], [32, 32]
    out = []
    for fuse in (True, False):
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE(sizes, gen, seed=11)
        model.build()
        tr = Trainer(model, head=LinkHead(sizes[-1], cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=b)
        assert tr._update_in_backward()
        tr.fuse_update = fuse
        step = tr.step_device if mode == 'device' else tr.step_graph
        losses = []
        for k in range(3):
            lk = np.asarray(links[k * b:(k + 1) * b])
            seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
            loss, _ = step(seeds, torch.from_numpy(targets[k * b:(k + 1) * b]).to(cuda),
                           torch.arange(2 * b, dtype=torch.int64, device=cuda) + 2 * b * k, rng_seed=9)
            losses.append(float(loss))
        out.append((losses, tr.opt.p_flat.clone(), tr.opt.m.clone(), tr.opt.v.clone(), tr.opt.t))
    (l0, p0, m0, v0, t0), (l1, p1, m1, v1, t1) = out
    assert t0 == t1 == 3
    assert np.allclose(l0, l1, rtol=1e-6, atol=0)
    assert float((p0 - p1).abs().max()) < 1e-6
    assert float((m0 - m1).abs().max()) <= 1e-6 * float(m1.abs().max())
    assert float((v0 - v1).abs().max()) <= 1e-6 * float(v1.abs().max())
    assert float((p0 - out[0][1]).abs().max()) == 0 and l0[2] < l0[0] + 1e-3


@pytest.mark.parametrize('ns,f,b', [([5, 3, 2], 64, 48), ([3, 10, 4], 32, 48), ([3, 2, 2], 64, 45)])     # (45 * 3 hop-1 nodes: a last group of 7)
def test_three_hop_backward_in_two_kernels_equals_the_layer_by_layer_launches(cuda, ns, f, b):
    """Three hops, 32 units per layer: a step with the forward pass as a kernel below the seeds and one over the seeds, the
    backward pass as a kernel over the seeds, a kernel over the hop-1 nodes and one reduction, against the layer-by-layer
    launches of either pass: same loss, same gradients up to the order of the sums; the two-kernel form repeats its bits, and with Adam inside its reduction the
    weights equal those of the separate update."""
    from graphsage_amd.train import LinkHead
    g, links, targets = _toy(5, fu=f, fi=f)
    sizes = [32, 32, 32]
    out = []
    for fused, fuse_update, subtree in ((True, True, True), (False, True, True), (True, False, True), (True, True, True),
                                        (True, True, False)):
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE(sizes, gen, seed=11)
        model.build()
        tr = Trainer(model, head=LinkHead(sizes[-1], cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=b)
        assert tr.engine.fused_backward3 and tr.engine.subtree_train and not tr.engine.tail_kernel
        tr.engine.fused_backward3 = fused
        tr.engine.subtree_train = subtree             # (False: the layer-by-layer forward pass under the two backward kernels)
        tr.fuse_update = fuse_update
        lk = np.asarray(links[:b])
        seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
        loss, pred = tr.step_device(seeds, torch.from_numpy(targets[:b]).to(cuda),
                                    torch.arange(2 * b, dtype=torch.int64, device=cuda), rng_seed=9)
        out.append((float(loss), tr.bucket.flat.clone(), tr.opt.p_flat.clone(), pred.clone()))
    scale = float(out[1][1].abs().max())
    assert scale > 0
    for other in (1, 4):                              # against the layered forward, under either backward pass
        assert abs(out[0][0] - out[other][0]) <= 1e-6 * max(1.0, abs(out[other][0]))
        assert float((out[0][3] - out[other][3]).abs().max()) < 1e-6
        assert float((out[0][1] - out[other][1]).abs().max()) <= 3e-6 * scale
    assert torch.equal(out[0][1], out[3][1]) and torch.equal(out[0][2], out[3][2]) and out[0][0] == out[3][0]
    assert torch.equal(out[0][1], out[2][1]) and torch.equal(out[0][2], out[2][2])      # Adam inside / after the reduction


def test_stage_batch_copies_inputs_and_rate_in_one_launch(cuda):
    from graphsage_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(1)
    for b in (1, 100, 5000):
        seeds = torch.randint(0, 1 << 30, (2 * b,), device=cuda, generator=gen, dtype=torch.int32)
        walks = torch.randint(0, 1 << 60, (2 * b,), device=cuda, generator=gen, dtype=torch.int64)
        y = torch.rand(b, device=cuda, generator=gen)
        d_seeds, d_walks, d_y = torch.zeros_like(seeds), torch.zeros_like(walks), torch.zeros_like(y)
        lr = torch.zeros(1, device=cuda)
        kernels.stage_batch(seeds, d_seeds, walks, d_walks, y, d_y, 0.125, lr)
        assert torch.equal(d_seeds, seeds) and torch.equal(d_walks, walks) and torch.equal(d_y, y)
        assert float(lr) == 0.125
    with pytest.raises(ValueError):
        kernels.stage_batch(seeds[:4], d_seeds, walks, d_walks, y, d_y)
    with pytest.raises(TypeError):
        kernels.stage_batch(seeds.long(), d_seeds, walks, d_walks, y, d_y)


@pytest.mark.parametrize('m,fs,fn,h,act', [(1040, 64, 64, 32, 'relu'), (4096 * 26, 64, 64, 32, 'relu'), (48, 32, 32, 32, 'linear'),
                                           (300, 24, 24, 32, 'relu'), (130, 128, 128, 64, 'linear')])
def test_paired_weight_gradients_equal_two_single_calls(cuda, m, fs, fn, h, act):
    """gs_sage_project_wgrad_pair (one launch for two problems; the (24, 24) case takes its
    fallback) against gs_sage_project_bwd on each problem."""
    from graphsage_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(m + fs)
    rnd = lambda *shape: torch.randn(shape, device=cuda, generator=gen)
    ws_floats = kernels.project_bwd_workspace_floats(fs, fn, h, m)
    problems, want = [], []
    for k in range(2):
        xs, xm, out, g = rnd(m, fs), rnd(m, fn), rnd(m, h), rnd(m, h)
        with_bias = k == 0
        want.append(kernels.sage_project_bwd(xs, xm, rnd(fs, h // 2), rnd(fn, h // 2), act, out, g,
                                             need_x_self=False, need_x_mean=False, need_bias=with_bias)[:3])
        problems.append({'x_self': xs, 'x_mean': xm, 'out': out if act == 'relu' or k else None, 'g_out': g,
                         'g_w_self': torch.full((fs, h // 2), float('nan'), device=cuda),
                         'g_w_neigh': torch.full((fn, h // 2), float('nan'), device=cuda),
                         'g_bias': torch.full((h,), float('nan'), device=cuda) if with_bias else None,
                         'workspace': torch.empty(max(ws_floats, 1), device=cuda)})
    call = kernels.WgradPairCall(problems, act)
    call.run()
    for p, (g_ws, g_wn, g_b) in zip(problems, want):
        assert torch.equal(p['g_w_self'], g_ws) and torch.equal(p['g_w_neigh'], g_wn)    # same kernels, same order of sums
        if p['g_bias'] is not None:
            assert torch.equal(p['g_bias'], g_b)


def test_forward_engine_built_before_the_trainer_follows_the_moved_weights(cuda):
    """A Trainer moves every parameter into one flat buffer.  A forward engine that ran before that keeps
    prepared calls holding the old addresses: it must notice and read the weights where they live now."""
    from graphsage_amd.engine import SageForwardEngine
    from graphsage_amd.train import LinkHead
    for f, ns in ((64, [6, 3]), (64, [4, 3, 2])):                  # one-kernel route; one kernel per layer route
        g, links, targets = _toy(4, fu=f, fi=f)
        b = 32
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE([32] * len(ns), gen, seed=11).build()
        lk = np.asarray(links[:b])
        seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
        wid = torch.arange(2 * b, dtype=torch.int64, device=cuda)
        t = torch.from_numpy(targets[:b]).to(cuda)
        eng = SageForwardEngine(g, model, b)
        before = [x.clone() for x in eng.forward(seeds, wid, rng_seed=9)]
        old_ptr = model.parameters()[0].data_ptr()
        tr = Trainer(model, head=LinkHead(32, cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=b)
        assert model.parameters()[0].data_ptr() != old_ptr         # the weights did move
        same = eng.forward(seeds, wid, rng_seed=9)
        assert torch.equal(same[0], before[0]) and torch.equal(same[1], before[1])
        for _ in range(3):
            tr.step_graph(seeds, t, wid, rng_seed=9)
        after = [x.clone() for x in eng.forward(seeds, wid, rng_seed=9)]
        assert not torch.equal(after[0], before[0])                # it sees the trained weights ...
        fresh = SageForwardEngine(g, model, b).forward(seeds, wid, rng_seed=9)
        assert torch.equal(after[0], fresh[0]) and torch.equal(after[1], fresh[1])   # ... exactly
        # and the model called on the same sampled batch agrees
        hops, _ = ga.kernels.sample(g.device_tensors()['indptr'], g.device_tensors()['indices'], seeds, ns, rng_seed=9,
                                    walk_ids=wid)
        slot_ids = [seeds[:b].reshape(b, 1), seeds[b:].reshape(b, 1)]
        for hop in hops:
            slot_ids += [hop[:b], hop[b:]]
        with torch.no_grad():
            emb = model(ga.BatchFeatures(g, b, slot_ids, gen._slot_types))
        for k in range(2):
            assert float((emb[k] - after[k]).abs().max()) <= 1e-5
        # a second Trainer on the same model moves the weights again: the first one notices, too
        tr2 = Trainer(model, head=tr.head, lr=1e-2, engine='plan', graph=g, batch_size=b)
        with pytest.raises(ga._lib.GsageError):
            tr.step_graph(seeds, t, wid, rng_seed=9)
        loss, _ = tr2.step_graph(seeds, t, wid, rng_seed=9)
        assert np.isfinite(float(loss))


def test_evaluate_and_predict_run_without_dropout(cuda):
    """Keras runs evaluate, predict and the validation pass of fit with dropout off; the model's training flag
    (dropout on by default) must not leak into them, and is restored afterwards."""
    g, links, targets = _toy(1, fu=32, fi=32)
    gen = ga.GraphSAGEGenerator(g, 16, [3, 2], head_node_types=['user', 'item'], seed=1)
    model = ga.GraphSAGE([32, 32], gen, dropout=0.5, seed=2)
    plain = ga.GraphSAGE([32, 32], gen, dropout=0.0, seed=2)
    from graphsage_amd.train import LinkHead
    tr = Trainer(model, head=LinkHead(32, cuda, seed=5), lr=1e-2, graph=g, batch_size=16)
    tr0 = Trainer(plain, head=LinkHead(32, cuda, seed=5), lr=1e-2, graph=g, batch_size=16)
    seq = gen.flow(links[:64], targets=targets[:64], shuffle=False)
    seq0 = gen.flow(links[:64], targets=targets[:64], shuffle=False)
    seq0.salt = seq.salt                                          # same draws for both
    assert model.training
    a, b2 = tr.evaluate(seq), tr.evaluate(seq)
    assert a == b2                                                # deterministic: no dropout mask was drawn
    assert model.training                                         # flag restored
    want = tr0.evaluate(seq0)
    assert abs(a['loss'] - want['loss']) < 1e-6
    p, p0 = tr.predict(seq), tr0.predict(seq0)
    assert float((p - p0).abs().max()) < 1e-6
    # training mode really drops: two passes through the model differ
    batch, _ = seq[0]
    x1, x2 = model(batch), model(batch)
    assert not torch.equal(x1[0], x2[0])


@pytest.mark.parametrize('ns', [[4, 3], [3, 3, 2]])
def test_predict_and_evaluate_with_two_batches_in_flight_equal_batch_by_batch_scoring(cuda, ns):
    """evaluate / predict keep two batches in flight on the one-kernel (two hops) and two-launch (three hops)
    routes: the scores are bit for bit those of scoring batch after batch, short last batch included."""
    from graphsage_amd.train import LinkHead
    g, links, targets = _toy(3, fu=64, fi=64)
    b = 16
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=1)
    model = ga.GraphSAGE([32] * len(ns), gen, seed=2)
    tr = Trainer(model, head=LinkHead(32, cuda, seed=5), lr=1e-2, graph=g, batch_size=b)
    n = 5 * b + 7                                                  # five full batches and a short one
    seq = gen.flow(links[:n], targets=targets[:n], shuffle=False)
    with torch.no_grad(), tr._inference():
        want = torch.cat([tr._scores(seq, i)[0] for i in range(len(seq))])
    assert tr._fwd_engine is not None and (tr._fwd_engine.walk_kernel or tr._fwd_engine.subtree_kernel)
    got = tr.predict(seq)
    assert tr._fwd_pipe is not None
    assert got.shape == (n, 1) and torch.equal(got, want)
    again = tr.predict(seq)                                        # the slots are reused
    assert torch.equal(again, want)
    ev = tr.evaluate(seq)
    y = torch.from_numpy(np.asarray(targets[:n], dtype=np.float32)).to(cuda)
    assert abs(ev['loss'] - float(((want.reshape(-1).double() - y.double()) ** 2).mean())) < 1e-6


def test_sequences_of_one_generator_draw_independently_and_leave_the_walker_seed_alone(cuda):
    g, links, targets = _toy(2, fu=32, fi=32)
    gen = ga.GraphSAGEGenerator(g, 16, [5, 3], head_node_types=['user', 'item'], seed=77)
    walker_seed = gen.sampler.seed
    first = [x.cpu() for x in gen.sample_features(links[:16], 0).ids]
    train = gen.flow(links[:64], targets=targets[:64], shuffle=False)
    test = gen.flow(links[:64], targets=targets[:64], shuffle=False)
    a, _ = train[0]
    c, _ = test[0]
    assert train.salt != test.salt and train.roots(0)[2] != test.roots(0)[2]
    assert not torch.equal(a.ids[4], c.ids[4])                    # same links, same positions: other neighborhoods
    a2, _ = train[0]
    assert all(torch.equal(x, y) for x, y in zip(a.ids, a2.ids))  # a sequence reproduces its own batches
    assert gen.sampler.seed == walker_seed                        # flow() batches did not re-key the walker
    gen2 = ga.GraphSAGEGenerator(g, 16, [5, 3], head_node_types=['user', 'item'], seed=77)
    again = [x.cpu() for x in gen2.sample_features(links[:16], 0).ids]
    assert all(torch.equal(x, y) for x, y in zip(first, again))   # running-counter calls follow the constructor seed


def test_adam_kernel_scales_the_gradients_it_reads(cuda):
    """grad_scale = 1 / world after a summing all-reduce equals dividing first."""
    from graphsage_amd import kernels
    gen = torch.Generator().manual_seed(3)
    n = 5000
    p0, gsum = torch.randn(n, generator=gen).to(cuda), (torch.randn(n, generator=gen) * 8).to(cuda)
    a = [p0.clone(), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)]
    c = [p0.clone(), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)]
    keep = gsum.clone()
    for t in range(1, 4):
        lr_t = 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        kernels.adam_step(a[0], gsum, a[1], a[2], lr_t, 0.9, 0.999, 1e-7, grad_scale=0.125)
        kernels.adam_step(c[0], gsum * 0.125, c[1], c[2], lr_t, 0.9, 0.999, 1e-7)
    assert torch.equal(gsum, keep)                                # the gradient buffer is only read
    for x, y in zip(a, c):
        assert torch.equal(x, y)                                  # 1/8 is exact in fp32: same bits


@pytest.mark.parametrize('b,n0,h1,h2,normalize,act', [(48, 8, 32, 32, True, 'linear'), (5, 25, 64, 64, True, 'linear'),
                                                      (130, 3, 32, 64, False, 'relu'), (1, 1, 64, 32, True, 'relu'),
                                                      (4096, 25, 32, 32, True, 'linear'), (9000, 2, 32, 32, True, 'linear')])
def test_tail_kernel_with_the_link_head_inside_equals_head_kernel_then_tail_kernel(cuda, b, n0, h1, h2, normalize, act):
    """gs_sage2_backward_tail_head + gs_link_head_reduce against gs_link_head_mse_fwd_bwd followed by
    gs_sage2_backward_tail: same predictions and loss, same head gradients (sums taken in another order:
    1e-5), same g_z / g_h1; and against the float64 oracle head."""
    from graphsage_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(7 * b + n0)
    rnd = lambda *shape: torch.randn(shape, device=cuda, generator=gen)
    types = ('user', 'item')
    z = {t: rnd(b, h2) for t in types}
    emb = {t: torch.nn.functional.normalize(z[t], dim=1) if normalize else z[t].clone() for t in types}
    ws = {t: rnd(h1, h2 // 2) * 0.2 for t in types}
    wn = {t: rnd(h1, h2 // 2) * 0.2 for t in types}
    k, bias = rnd(2 * h2) * 0.3, rnd(1) * 0.1
    y = (torch.rand(b, device=cuda, generator=gen) > 0.5).float()
    # the two-launch route
    gk0, gb0 = torch.zeros(2 * h2, device=cuda), torch.zeros(1, device=cuda)
    step = kernels.LinkHeadStep(b, h2, k, bias, gk0, gb0, cuda)
    loss0, pred0 = step.run(emb['user'], emb['item'], y)
    g_z0 = {t: torch.empty(b, h2, device=cuda) for t in types}
    g_h0 = {t: torch.empty(b * (1 + n0), h1, device=cuda) for t in types}
    kernels.sage2_backward_tail(z, {'user': step.g_user, 'item': step.g_item}, ws, wn, n0, normalize, act, g_z0, g_h0)
    # the one-launch route
    gk1, gb1 = torch.full((2 * h2,), float('nan'), device=cuda), torch.full((1,), float('nan'), device=cuda)
    loss1, pred1 = torch.full((1,), float('nan'), device=cuda), torch.full((b,), float('nan'), device=cuda)
    g_z1 = {t: torch.full((b, h2), float('nan'), device=cuda) for t in types}
    g_h1 = {t: torch.full((b * (1 + n0), h1), float('nan'), device=cuda) for t in types}
    slab = kernels.tail_head_slab(b, h2, cuda)
    slab.fill_(float('nan'))
    kernels.sage2_backward_tail_head(z, emb, y, k, bias, ws, wn, n0, normalize, act, g_z1, g_h1, pred1, slab)
    kernels.link_head_reduce(slab, b, h2, gk1, gb1, loss1)
    assert float((pred1 - pred0).abs().max()) <= 2e-7             # (32-wide rows: the two kernels add the 64 products in another order)
    assert abs(float(loss1) - float(loss0)) <= 1e-6 * max(1.0, float(loss0))
    for got, want in ((gk1, gk0), (gb1, gb0)):
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-9
    for t in types:
        for got, want in ((g_z1[t], g_z0[t]), (g_h1[t], g_h0[t])):
            assert float((got - want).abs().max()) <= 2e-6 * max(float(want.abs().max()), 1e-12)
    # float64 head
    ku = torch.cat([emb['user'], emb['item']], 1).double()
    pr = torch.sigmoid(ku @ k.double() + bias.double())
    assert float((pred1.double() - pr).abs().max()) < 1e-6
    gz = 2 * (pr - y.double()) / b * pr * (1 - pr)
    want_gk = (ku * gz[:, None]).sum(0)
    assert float((gk1.double() - want_gk).abs().max()) <= 1e-5 * float(want_gk.abs().max()) + 1e-9
    assert abs(float(gb1) - float(gz.sum())) <= 1e-5 * abs(float(gz.sum())) + 1e-9


@pytest.mark.parametrize('m,fs,fn,h,act', [(1040, 64, 64, 32, 'relu'), (4096 * 26, 64, 64, 32, 'relu'), (50, 32, 32, 32, 'linear'),
                                           (777, 128, 128, 64, 'relu'), (3, 64, 64, 64, 'linear')])
def test_weight_gradients_with_own_rows_read_by_id_equal_the_gathered_form(cuda, m, fs, fn, h, act):
    """gs_sage_project_wgrad_pair with self_ids (own rows taken from the feature table inside the kernel)
    against the same call on gathered copies of those rows; ids outside the table count as zero rows."""
    from graphsage_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(m + fs)
    rnd = lambda *shape: torch.randn(shape, device=cuda, generator=gen)
    assert kernels.load().gs_sage_project_wgrad_by_id_supported(fs, fn, h)
    ws_floats = kernels.project_bwd_workspace_floats(fs, fn, h, m)
    rows, off = 5000, 1000
    by_id, gathered = [], []
    for kk in range(2):
        table = rnd(rows, fs)
        ids = torch.randint(off, off + rows, (m,), device=cuda, generator=gen, dtype=torch.int32)
        ids[::7] = -1                                            # below a node without neighbors
        if m > 2:
            ids[1], ids[2] = off + rows, off - 1                 # just outside either end
        xs = kernels.gather_rows(table, ids, off)
        xm, out, g = rnd(m, fn), rnd(m, h), rnd(m, h)
        mk = lambda: {'x_mean': xm, 'out': out if act == 'relu' else None, 'g_out': g,
                      'g_w_self': torch.full((fs, h // 2), float('nan'), device=cuda),
                      'g_w_neigh': torch.full((fn, h // 2), float('nan'), device=cuda),
                      'g_bias': torch.full((h,), float('nan'), device=cuda),
                      'workspace': torch.empty(max(ws_floats, 1), device=cuda)}
        by_id.append(dict(mk(), x_self=table, self_ids=ids, self_off=off))
        gathered.append(dict(mk(), x_self=xs))
    kernels.WgradPairCall(by_id, act).run()
    kernels.WgradPairCall(gathered, act).run()
    for p, q in zip(by_id, gathered):
        for name in ('g_w_self', 'g_w_neigh', 'g_bias'):
            assert torch.equal(p[name], q[name]), name             # same rows, same order of sums: same bits
    assert not kernels.load().gs_sage_project_wgrad_by_id_supported(24, 24, 32)
    with pytest.raises(Exception):
        bad = [dict(p, x_self=torch.zeros(10, 24, device=cuda), x_mean=torch.zeros(m, 24, device=cuda)) for p in by_id]
        kernels.WgradPairCall(bad, act)


@pytest.mark.parametrize('ns,f,b', [([5, 3, 2], 64, 48), ([3, 2, 2], 32, 45), ([25, 10, 5], 64, 24)])
def test_saving_subtree_forward_leaves_the_rows_of_the_layered_forward(cuda, ns, f, b):
    """What the two-launch training forward (subtree kernel + seeds kernel, saving forms) leaves for the backward pass
    against the layer-by-layer forward on the same draws: ids exact, means and outputs to rounding, the hop-2 mask words
    against the sign of the layered outputs."""
    from graphsage_amd.train import LinkHead
    g, links, _ = _toy(5, fu=f, fi=f)
    n0, n1, _ = ns
    lk = np.asarray(links[:b])
    seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
    walks = torch.arange(2 * b, dtype=torch.int64, device=cuda) + 1000
    engines = []
    for subtree in (True, False):
        gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
        model = ga.GraphSAGE([32, 32, 32], gen, seed=11)
        model.build()
        tr = Trainer(model, head=LinkHead(32, cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=b)
        e = tr.engine
        assert e.subtree_train
        e.subtree_train = subtree
        emb = e.forward_train(seeds, walks, rng_seed=21)
        torch.cuda.synchronize()
        engines.append((tr, e, [x.clone() for x in emb]))
    (_, sub, emb_s), (_, lay, emb_l) = engines
    close = lambda x, y, what: (float((x - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max()))) or pytest.fail(what)
    for x, y in zip(emb_s, emb_l):
        close(x, y, 'embeddings')
    r1, r2 = b * (1 + n0), b * (1 + n0 + n0 * n1)
    st = sub._sub_train
    # the drawn ids: hops 1 and 2 as the sampler lays them out (user-rooted walks first)
    assert torch.equal(st['hop1'], lay.hops[0].reshape(-1)) and torch.equal(st['hop2'], lay.hops[1].reshape(-1))
    for t, first in (('user', False), ('item', True)):               # hop-1 nodes of type t: items hang below the user seeds (first half)
        half = slice(0, b * n0) if first else slice(b * n0, 2 * b * n0)
        close(st['h0'][half], lay.h[0][t][b:r1], 'h0 of hop-1 nodes, ' + t)
        close(st['h1'][half], lay.h[1][t][b:r1], 'h1 of hop-1 nodes, ' + t)
    for t in ('user', 'item'):
        close(sub.x_mean[0][t][:r2], lay.x_mean[0][t][:r2], 'layer-0 neighbor means, ' + t)
        close(sub.x_mean[1][t][:r1], lay.x_mean[1][t][:r1], 'layer-1 neighbor means, ' + t)
        close(sub.x_mean[2][t][:b], lay.x_mean[2][t][:b], 'layer-2 neighbor means, ' + t)
        close(sub.h[0][t][:b], lay.h[0][t][:b], 'h0 of seeds, ' + t)
        close(sub.h[1][t][:b], lay.h[1][t][:b], 'h1 of seeds, ' + t)
        close(sub.z[t], lay.z[t], 'z, ' + t)
        words = st['mask'][t].cpu().numpy().view(np.uint32)
        bits = ((words[:, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
        rows = lay.h[0][t][r1:r2].cpu().numpy()
        wrong = bits != (rows > 0)
        assert not (wrong & (rows > 1e-6)).any()                        # (an output that rounds to zero on one route may be a tiny positive on the other)
        assert wrong.mean() < 1e-3


@pytest.mark.parametrize('ns,f,b', [([5, 3, 2], 64, 48), ([25, 10, 5], 64, 21)])
def test_three_hop_training_forward_gives_the_bits_of_the_inference_forward(cuda, ns, f, b):
    """Saving what the backward pass reads changes nothing in what is computed: the two-launch forward of the training
    plan (saving forms) and of the forward engine (plain forms) on the same draws, bit for bit."""
    from graphsage_amd.engine import SageForwardEngine
    from graphsage_amd.train import LinkHead
    g, links, _ = _toy(5, fu=f, fi=f)
    lk = np.asarray(links[:b])
    seeds = torch.from_numpy(np.concatenate([lk[:, 0], lk[:, 1]]).astype(np.int32)).to(cuda)
    walks = torch.arange(2 * b, dtype=torch.int64, device=cuda) + 77
    gen = ga.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'], seed=3)
    model = ga.GraphSAGE([32, 32, 32], gen, seed=11)
    model.build()
    tr = Trainer(model, head=LinkHead(32, cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=b)
    assert tr.engine.subtree_train
    train_emb = [x.clone() for x in tr.engine.forward_train(seeds, walks, rng_seed=4)]
    fwd = SageForwardEngine(g, model, b)
    assert fwd.subtree_kernel
    infer_emb = fwd.forward(seeds, walks, rng_seed=4)
    for x, y in zip(train_emb, infer_emb):
        assert torch.equal(x, y)
