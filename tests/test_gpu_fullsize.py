"""Parity at the benchmark's full size (1M users x 100k items, 50M edges,
fan-outs [25,10], 4096 seed edges per batch) and at sizes past 32-bit offsets --
needs a GPU.

At these sizes the oracle's Python-loop forms are out of reach, so the checks
are: the numpy-vectorised oracle walk for every root of a full batch (bit
exact), the float64 oracle model on a subset of the batch, and size-independent
properties: two independent kernel paths agree, a link's embedding does not
depend on the batch it travels in, embeddings have unit length.
"""
import numpy as np
import pytest
import torch

import graphsage_amd as ga
from graphsage_amd import kernels, synthetic
from graphsage_amd.engine import SageForwardEngine
from oracle import generator as ogen
from oracle import model as omodel
from oracle import walker

pytestmark = pytest.mark.gpu

NU, NI, E, F, FAN, B = 1_000_000, 100_000, 50_000_000, 64, [25, 10], 4096
SEED = 0x9E3779B97F4A7C15


@pytest.fixture(scope='module')
def big(cuda):
    g, u, it = synthetic.power_law_bipartite(NU, NI, E, F, F, alpha_item=1.1, seed=1234, device=cuda)
    hu, hi = synthetic.seed_edges(u, it, 1 << 20, seed=0)
    gen = ga.GraphSAGEGenerator(g, B, FAN, head_node_types=['user', 'item'], seed=1)
    model = ga.GraphSAGE([32, 32], gen, seed=11).build()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.1, 0.1)
    return dict(g=g, u=u, it=it, hu=hu, hi=hi, model=model, dev=g.device_tensors())


def test_adjacency_of_100m_entries_equals_a_stable_sort_by_source(big, cuda):
    dev = big['dev']
    hu = big['u'].to(torch.int64)
    hi = big['it'].to(torch.int64)
    src = torch.cat([hu, hi])
    dst = torch.cat([big['it'], big['u']])
    order = torch.sort(src, stable=True).indices
    assert torch.equal(dev['indices'], dst[order])
    deg = torch.bincount(src, minlength=NU + NI)
    assert torch.equal(dev['indptr'][1:], torch.cumsum(deg, 0)) and int(dev['indptr'][0]) == 0
    assert int(deg.min()) >= 1


def test_full_batch_walks_equal_the_oracle_bit_for_bit(big, cuda):
    dev = big['dev']
    seeds = torch.cat([big['hu'][:B], big['hi'][:B]]).contiguous()
    base = 3 * (1 << 40) + 17
    hops, err = kernels.sample(dev['indptr'], dev['indices'], seeds, FAN, rng_seed=SEED, walk_id_base=base)
    assert err.tolist() == [0, 0]
    indptr, indices = dev['indptr'].cpu().numpy(), dev['indices'].cpu().numpy()
    assert indices.shape[0] == 2 * E
    want = walker.walk_vectorised(indptr, indices, seeds.cpu().numpy(), FAN, rng_seed=SEED, first_walk=base)
    for got, w in zip(hops, want):
        np.testing.assert_array_equal(got.cpu().numpy(), w)
    # node types alternate with the hop: user roots -> items -> users
    h1, h2 = hops
    assert (h1[:B] >= NU).all() and (h1[B:] < NU).all() and (h2[:B] < NU).all() and (h2[B:] >= NU).all()


def test_one_kernel_forward_agrees_with_layered_path_and_oracle_subset(big, cuda):
    g, model = big['g'], big['model']
    seeds = torch.cat([big['hu'][B:2 * B], big['hi'][B:2 * B]]).contiguous()
    walk_ids = torch.cat([2 * torch.arange(B, 2 * B, dtype=torch.int64, device=cuda),
                          2 * torch.arange(B, 2 * B, dtype=torch.int64, device=cuda) + 1])
    eng_w = SageForwardEngine(g, model, B)
    eng_l = SageForwardEngine(g, model, B, walk_kernel=False, keep_ids=True, draw_in_layer=True)
    eng_u = SageForwardEngine(g, model, B, use_fused=False)
    assert eng_w.walk_kernel and all(eng_l.fused_layers)
    out_w = [x.clone() for x in eng_w.forward(seeds, walk_ids, rng_seed=SEED)]
    out_l = [x.clone() for x in eng_l.forward(seeds, walk_ids, rng_seed=SEED)]
    out_u = [x.clone() for x in eng_u.forward(seeds, walk_ids, rng_seed=SEED)]
    assert eng_w.err_flag.tolist() == [0, 0] and eng_l.err_flag.tolist() == [0, 0]
    for k in range(2):
        assert torch.isfinite(out_w[k]).all()
        assert (out_w[k] - out_l[k]).abs().max().item() <= 1e-5          # unit-length rows: scale 1
        assert (out_w[k] - out_u[k]).abs().max().item() <= 1e-5
        assert (out_w[k].double().norm(dim=1) - 1).abs().max().item() <= 1e-5

    # float64 oracle on the first links of the batch, from the ids the layered path sampled
    n = 48
    hops = [h.cpu().numpy().astype(np.int64) for h in eng_l.hops]
    sub = [np.concatenate([h[:n], h[B:B + n]]) for h in hops]
    links = [(int(a), int(b)) for a, b in zip(seeds[:n].tolist(), seeds[B:B + n].tolist())]
    feats = {'user': big['dev']['user'].cpu().numpy(), 'item': big['dev']['item'].cpu().numpy()}
    xin = ogen.sample_features_fast(feats, NU, links, sub)
    plan = omodel.LayerPlan([32, 32], FAN, g.node_feature_sizes)
    weights = {(l, t): {'w_self': a.w_self.detach().cpu().double(), 'w_neigh': a.w_neigh.detach().cpu().double(),
                        'bias': a.bias.detach().cpu().double()}
               for l, per in enumerate(model._aggs) for t, a in per.items()}
    want = omodel.forward(plan, weights, [torch.from_numpy(np.asarray(x, dtype=np.float64)) for x in xin])
    for k in range(2):
        assert (out_w[k][:n].cpu().double() - want[k]).abs().max().item() <= 1e-5


def test_embedding_of_a_link_does_not_depend_on_its_batch(big, cuda):
    g, model = big['g'], big['model']
    hu, hi = big['hu'], big['hi']
    ids = lambda lo, hi_: torch.cat([2 * torch.arange(lo, hi_, dtype=torch.int64, device=cuda),
                                     2 * torch.arange(lo, hi_, dtype=torch.int64, device=cuda) + 1])
    eng = SageForwardEngine(g, model, B)
    full = [x.clone() for x in eng.forward(torch.cat([hu[:B], hi[:B]]).contiguous(), ids(0, B), rng_seed=5)]
    small = SageForwardEngine(g, model, 1000)
    lo = 1536
    part = small.forward(torch.cat([hu[lo:lo + 1000], hi[lo:lo + 1000]]).contiguous(), ids(lo, lo + 1000), rng_seed=5)
    assert torch.equal(part[0], full[0][lo:lo + 1000]) and torch.equal(part[1], full[1][lo:lo + 1000])
    # another sampling key gives other neighborhoods
    other = eng.forward(torch.cat([hu[:B], hi[:B]]).contiguous(), ids(0, B), rng_seed=6)
    assert not torch.equal(other[0], full[0])


class _DeviceIndices:
    """Neighbor array left on the GPU; the oracle fetches the entries it needs."""
    lazy = True

    def __init__(self, t):
        self.t = t

    def __getitem__(self, pos):
        p = torch.from_numpy(np.ascontiguousarray(pos)).to(self.t.device)
        return self.t[p].cpu().numpy()


def test_sampler_addresses_neighbor_arrays_beyond_32_bit_offsets(cuda):
    """2 000 nodes with ~2.2 M neighbors each: 4.4e9 CSR entries (17.6 GB), so
    row starts pass 2**32 and degrees pass 2**21."""
    if torch.cuda.get_device_properties(0).total_memory < 60 * 2 ** 30:
        pytest.skip("needs about 20 GB of device memory")
    n_half, deg = 1000, 2_200_003
    total = 2 * n_half * deg
    assert total > 2 ** 32
    indices = torch.empty(total, dtype=torch.int32, device=cuda)
    gen = torch.Generator(device=cuda)
    gen.manual_seed(9)
    chunk = 1 << 28
    half = n_half * deg                                    # users' neighbors are items and vice versa
    for a0, a1, lo, hi in ((0, half, n_half, 2 * n_half), (half, total, 0, n_half)):
        for a in range(a0, a1, chunk):
            n = min(chunk, a1 - a)
            indices[a:a + n] = torch.randint(lo, hi, (n,), generator=gen, device=cuda, dtype=torch.int32)
    indptr = torch.arange(2 * n_half + 1, dtype=torch.int64, device=cuda) * deg
    roots = torch.tensor([0, 999, 1000, 1999, 1955, 1954, 7, 1500], dtype=torch.int32, device=cuda)
    hops, err = kernels.sample(indptr, indices, roots, [25, 10], rng_seed=77, walk_id_base=(1 << 50))
    assert err.tolist() == [0, 0]
    want = walker.walk_vectorised(indptr.cpu().numpy(), _DeviceIndices(indices), roots.cpu().numpy(), [25, 10],
                                  rng_seed=77, first_walk=(1 << 50))
    for got, w in zip(hops, want):
        np.testing.assert_array_equal(got.cpu().numpy(), w)
    # the positions drawn really lie past 2**32 for the late rows
    assert int(indptr[1954]) > 2 ** 32
    del indices
    torch.cuda.empty_cache()


def test_walk_kernel_reads_feature_tables_beyond_4_gib_with_plain_loads(cuda):
    """A user table of 17.5 M rows x 64 floats (4.5 GB) is past what a buffer
    descriptor addresses, so the one-kernel forward takes its 64-bit load path;
    it must agree with the general kernels."""
    if torch.cuda.get_device_properties(0).total_memory < 40 * 2 ** 30:
        pytest.skip("needs about 12 GB of device memory")
    nu, ni, e, b = 17_500_000, 50_000, 40_000_000, 512
    g, u, it = synthetic.power_law_bipartite(nu, ni, e, 64, 64, alpha_item=1.0, seed=5, device=cuda)
    assert g.device_tensors()['user'].numel() * 4 > 2 ** 32
    gen = ga.GraphSAGEGenerator(g, b, [25, 10], head_node_types=['user', 'item'], seed=1)
    model = ga.GraphSAGE([32, 32], gen, seed=3).build()
    # roots from the top of the id range: their rows lie beyond the 4 GiB mark
    pick = torch.arange(nu - b, nu, device=cuda)               # the per-user forced edges of the last users
    seeds = torch.cat([u[pick], it[pick]]).contiguous()
    assert int(seeds[:b].min()) * 256 > 2 ** 32
    walk_ids = torch.arange(2 * b, dtype=torch.int64, device=cuda)
    eng_w = SageForwardEngine(g, model, b)
    eng_u = SageForwardEngine(g, model, b, use_fused=False)
    assert eng_w.walk_kernel
    out_w = [x.clone() for x in eng_w.forward(seeds, walk_ids, rng_seed=3)]
    out_u = [x.clone() for x in eng_u.forward(seeds, walk_ids, rng_seed=3)]
    assert eng_w.err_flag.tolist() == [0, 0]
    for k in range(2):
        assert (out_w[k] - out_u[k]).abs().max().item() <= 1e-5
    del g, u, it, eng_w, eng_u
    torch.cuda.empty_cache()


def test_wide_layer_kernel_reads_a_table_beyond_4_gib(cuda):
    """A 4.5 GB neighbor table: the wide fused layer keeps row indices (not 32-bit byte offsets) in LDS and
    forms 64-bit addresses.  Checked against gather-mean + projection on the same ids, and a few rows against
    float64 arithmetic on the host."""
    rows, fn, fs, h = 1_100_000, 1024, 64, 256
    gen = torch.Generator(device=cuda).manual_seed(3)
    t_neigh = torch.randn((rows, fn), device=cuda, generator=gen)
    assert t_neigh.numel() * 4 > 1 << 32
    t_self = torch.randn((5000, fs), device=cuda, generator=gen)
    ws = torch.randn((fs, h // 2), device=cuda, generator=gen) * 0.1
    wn = torch.randn((fn, h // 2), device=cuda, generator=gen) * 0.03
    bias = torch.randn(h, device=cuda, generator=gen) * 0.1
    segs, want = [], []
    for n_heads, nc in [(3000, 5), (300, 12)]:
        heads = torch.randint(0, t_self.shape[0], (n_heads,), device=cuda, generator=gen, dtype=torch.int32)
        kids = torch.randint(0, rows, (n_heads * nc,), device=cuda, generator=gen, dtype=torch.int32)
        kids[-nc:] = rows - 1                                  # the last row of the table: highest address
        kids[::53] = -1
        segs.append({'head_ids': heads, 'child_ids': kids, 'n_heads': n_heads, 'n_child': nc})
        xs = kernels.gather_rows(t_self, heads, 0)
        xm = kernels.gather_mean(t_neigh, kids, n_heads, nc, 0)
        want.append(kernels.sage_project(xs, xm, ws, wn, bias, 'relu'))
    want = torch.cat(want)
    assert kernels.fused_supported(fs, fn, h, 12)
    got = kernels.sage_fused(t_self, 0, t_neigh, 0, segs, ws, wn, bias, 'relu')
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * scale
    # last head of the second segment on the host in float64
    kid = segs[1]['child_ids'][-12:].cpu().numpy()
    xn = np.stack([t_neigh[k].cpu().numpy().astype(np.float64) if k >= 0 else np.zeros(fn) for k in kid]).mean(0)
    xs = t_self[int(segs[1]['head_ids'][-1])].cpu().numpy().astype(np.float64)
    row = np.concatenate([xs @ ws.cpu().numpy().astype(np.float64), xn @ wn.cpu().numpy().astype(np.float64)])
    row = np.maximum(row + bias.cpu().numpy(), 0.0)
    assert np.abs(got[-1].cpu().numpy() - row).max() <= 2e-5 * max(1.0, np.abs(row).max())


def test_three_hop_training_step_at_full_batch_two_forward_routes_agree(big, cuda):
    """B = 4096, fan-outs [25, 10, 5] on the benchmark graph: a training step with the forward pass as the saving
    subtree kernel + seeds kernel against the same step on the layer-by-layer forward (both under the two-kernel
    backward pass): same loss and gradients up to the order of the sums; the step repeats its bits."""
    from graphsage_amd.train import LinkHead, Trainer
    g = big['g']
    fan = [25, 10, 5]
    seeds = torch.cat([big['hu'][:B], big['hi'][:B]]).contiguous()
    walks = torch.arange(2 * B, dtype=torch.int64, device=cuda) + (1 << 35)
    targets = (torch.arange(B, device=cuda) % 2).float()
    out = []
    for subtree in (True, False, True):
        gen = ga.GraphSAGEGenerator(g, B, fan, head_node_types=['user', 'item'], seed=1)
        model = ga.GraphSAGE([32, 32, 32], gen, seed=11).build()
        tr = Trainer(model, head=LinkHead(32, cuda, seed=5), lr=1e-2, engine='plan', graph=g, batch_size=B)
        assert tr.engine.subtree_train and tr.engine.fused_backward3
        tr.engine.subtree_train = subtree
        loss, pred = tr.step_device(seeds, targets, walks, rng_seed=SEED)
        assert tr.engine.err_flag.tolist() == [0, 0]
        out.append((float(loss), tr.bucket.flat.clone(), pred.clone(), tr.opt.p_flat.clone()))
        del tr, model, gen
        torch.cuda.empty_cache()
    assert np.isfinite(out[0][0]) and torch.isfinite(out[0][1]).all()
    scale = float(out[1][1].abs().max())
    assert scale > 0
    assert abs(out[0][0] - out[1][0]) <= 1e-6 * max(1.0, abs(out[1][0]))
    assert float((out[0][2] - out[1][2]).abs().max()) <= 2e-6
    assert float((out[0][1] - out[1][1]).abs().max()) <= 2e-5 * scale
    assert out[0][0] == out[2][0] and torch.equal(out[0][1], out[2][1]) and torch.equal(out[0][3], out[2][3])
