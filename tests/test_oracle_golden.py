"""The CPU oracle against the vectors the reference itself produced (no GPU).

Everything the GPU parity tests later compare against is pinned here first:
adjacency order, the walk for a given stream of pre-drawn words, the slot
regroup/interlace, the feature gather, batching and shuffling.
"""
import random

import numpy as np
import pytest

from conftest import golden_graph_parts
from oracle import generator as ogen
from oracle import philox, walker
from oracle.graph import OracleGraph


def _kat_graph(arrays):
    feats = {'user': arrays['kat_feat_user'], 'item': arrays['kat_feat_item']}
    return OracleGraph(feats, {'user': [0, 1, 2], 'item': [3, 4]}, arrays['kat_edges'])


def test_neighbor_order_is_edge_column_order(golden):
    arrays, meta = golden
    g = _kat_graph(arrays)
    assert {str(k): v for k, v in g.neighbor_dict.items()} == meta['kat_neighbor_dict']


def test_walk_known_answer_with_stdlib_stream(golden):
    arrays, meta = golden
    g = _kat_graph(arrays)
    random.seed(7)
    assert walker.walk_sequential(g, [3, 2], [0, 4], random) == meta['kat_walk_seed7']


def test_choices_is_floor_of_uniform_times_degree(golden):
    """random.choices(pop, k) consumes one random() per draw and picks
    pop[floor(u * len(pop))]: a walk is a function of the uniform stream alone."""
    arrays, meta = golden

    class FromUniforms:
        def __init__(self, us):
            self.us, self.pos = us, 0

        def choices(self, pop, k=1):
            out = [pop[int(self.us[self.pos + i] * len(pop))] for i in range(k)]
            self.pos += k
            return out

    g = _kat_graph(arrays)
    got = walker.walk_sequential(g, [3, 2], [0, 4], FromUniforms(meta['kat_uniforms_seed7']))
    assert got == meta['kat_walk_seed7']


def test_generator_known_answer(golden):
    arrays, meta = golden
    g = _kat_graph(arrays)
    random.seed(3)
    seq = ogen.OracleLinkSequence(
        lambda heads, bn: ogen.sample_features_faithful(g, [3, 2], heads, random),
        2, [(0, 3), (2, 4), (1, 3)], targets=[1, 0, 1], shuffle=False)
    feats, targets = seq[0]
    assert len(seq) == meta['kat_gen_seed3_len']
    assert [list(a.shape) for a in feats] == meta['kat_gen_seed3_shapes']
    assert [int(t) for t in targets] == meta['kat_gen_seed3_targets']
    for i, a in enumerate(feats):
        np.testing.assert_array_equal(a, arrays['kat_gen_seed3_x%d' % i])


def test_schema_literals_for_two_hops(golden):
    _, meta = golden
    assert [[t, l] for t, l in ogen.type_adjacency_list(2)] == meta['kat_type_adjacency_list']
    got = [[[t, list(l)] for t, l in row] for row in ogen.sampling_schema([3, 2])]
    assert got == meta['kat_sampling_schema']


@pytest.mark.parametrize('case', range(7))
def test_tape_replay_walks_and_features(golden, case):
    arrays, meta = golden
    info = meta['replay'][case]
    key, ns, b = info['key'], info['num_samples'], info['batch']
    g = OracleGraph(*golden_graph_parts(arrays, info['graph']))
    links = [tuple(int(v) for v in l) for l in arrays[key + '_links']]
    tape = arrays[key + '_tape']
    roots = [l[0] for l in links] + [l[1] for l in links]

    # sequential (reference-shaped) walk == keyed walk == vectorised walk == reference
    seq_walks = walker.walk_sequential(g, ns, roots, walker.TapeRS(tape))
    hops_seq = walker.walks_to_hops(seq_walks, ns)
    hops_key = walker.walk_keyed(g, ns, roots, walker.tape_draw(tape, ns))
    order = np.argsort(np.asarray(g.edges[0]), kind='stable')
    n = g.correction_constant + g.node_features['item'].shape[0]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(g.edges[0], minlength=n), out=indptr[1:])
    indices = np.asarray(g.edges[1])[order]
    hops_vec = walker.walk_vectorised(indptr, indices, roots, ns, tape=tape)
    for h in range(len(ns)):
        ref = arrays[key + '_hop%d' % (h + 1)]
        np.testing.assert_array_equal(hops_seq[h], ref)
        np.testing.assert_array_equal(hops_key[h], ref)
        np.testing.assert_array_equal(hops_vec[h], ref)

    # features: faithful regroup and the dense-array form
    feats = ogen.sample_features_faithful(g, ns, links, walker.TapeRS(tape))
    fast = ogen.sample_features_fast(g.node_features, info['num_users'], links, hops_vec)
    assert len(feats) == 2 * (len(ns) + 1)
    for i, a in enumerate(feats):
        np.testing.assert_array_equal(a, arrays[key + '_x%d' % i])
        np.testing.assert_array_equal(fast[i], arrays[key + '_x%d' % i])


@pytest.mark.parametrize('case', range(3))
def test_tape_replay_walker_only_including_three_hops(golden, case):
    arrays, meta = golden
    info = meta['walker_only'][case]
    key, ns = info['key'], info['num_samples']
    g = OracleGraph(*golden_graph_parts(arrays, info['graph']))
    roots = [int(r) for r in arrays[key + '_roots']]
    tape = arrays[key + '_tape']
    walks = walker.walk_sequential(g, ns, roots, walker.TapeRS(tape))
    hops = walker.walks_to_hops(walks, ns)
    keyed = walker.walk_keyed(g, ns, roots, walker.tape_draw(tape, ns))
    for h in range(len(ns)):
        np.testing.assert_array_equal(hops[h], arrays[key + '_hop%d' % (h + 1)])
        np.testing.assert_array_equal(keyed[h], arrays[key + '_hop%d' % (h + 1)])
    if case == 0:
        assert walks == meta['walk0_nested']
        assert walker.hops_to_walks(roots, hops, ns) == meta['walk0_nested']


def test_empty_neighbor_list_raises_like_the_reference(golden):
    _, meta = golden
    edges = np.array([[0, 1, 3, 3], [3, 3, 0, 1]])
    feats = {'user': np.zeros((3, 2)), 'item': np.zeros((2, 2))}
    g = OracleGraph(feats, {'user': [0, 1, 2], 'item': [3, 4]}, edges)
    assert meta['errors']['no_out_edges'] == 'KeyError'
    with pytest.raises(KeyError):
        walker.walk_sequential(g, [2, 2], [2], random)
    g.neighbor_dict[2] = []
    with pytest.raises(ValueError) as ei:
        walker.walk_sequential(g, [2, 2], [2], random)
    assert [type(ei.value).__name__, str(ei.value)] == meta['errors']['empty_list']


def test_link_sequence_order_and_short_last_batch(golden):
    _, meta = golden
    exp = meta['sequence']
    calls = []

    def sample(heads, bn):
        calls.append([list(h) for h in heads])
        return None

    ids = [(i, 100 + i) for i in range(23)]
    targets = [i % 2 for i in range(23)]
    seq = ogen.OracleLinkSequence(sample, 5, ids, targets=targets, shuffle=True, seed=0)
    assert len(seq) == exp['len']
    assert seq.indices == exp['epoch0']
    _, t = seq[4]
    assert [int(v) for v in t] == exp['last_batch_targets']
    assert calls[-1] == exp['last_batch_heads']
    seq.on_epoch_end()
    assert seq.indices == exp['epoch1']
    assert ogen.OracleLinkSequence(sample, 5, ids, targets=targets, shuffle=False, seed=0).indices == exp['noshuffle']


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    f = lambda t: [int(x) for x in t]
    assert f(philox.philox4x32_10(0, 0, 0, 0, 0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert f(philox.philox4x32_10(*[0xffffffff] * 6)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert f(philox.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_integer_index_equals_float_floor_below_2_to_21():
    """floor(u * n) with u = r / 2**32 equals (r * n) >> 32 while the float product
    is exact; the integer form is what both sides use, valid for any degree."""
    rng = np.random.default_rng(0)
    r = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64)
    n = rng.integers(1, 2 ** 20, 20000, dtype=np.uint64)
    assert np.array_equal(philox.index_from_u32(r, n), np.floor(r.astype(np.float64) / 2.0 ** 32 * n).astype(np.int64))


def test_keyed_and_vectorised_philox_walks_agree(golden):
    arrays, _ = golden
    g = OracleGraph(*golden_graph_parts(arrays, 'g_small'))
    n = g.correction_constant + g.node_features['item'].shape[0]
    order = np.argsort(np.asarray(g.edges[0]), kind='stable')
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(g.edges[0], minlength=n), out=indptr[1:])
    indices = np.asarray(g.edges[1])[order]
    roots = [0, 5, 41, 39, 51]
    a = walker.walk_keyed(g, [4, 3, 2], roots, walker.philox_draw(1234), first_walk=77)
    b = walker.walk_vectorised(indptr, indices, roots, [4, 3, 2], rng_seed=1234, first_walk=77)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


# ------------------------------------------- the cyclic walker (backup/sampler_backup.py) ----

def _split_perms(arrays, key):
    lengths = arrays[key + '_perm_lengths']
    flat = arrays[key + '_perms']
    out, pos = [], 0
    for n in lengths:
        out.append(flat[pos:pos + int(n)])
        pos += int(n)
    return out


@pytest.mark.parametrize('case', range(4))
def test_cyclic_walk_restatement_equals_the_reference(golden, case):
    """oracle.cyclic.walk_cyclic on the permutations the reference's own walker was given: same walks,
    call after call, and the same cursors at the end."""
    from oracle import cyclic
    arrays, meta = golden
    info = meta['cyclic'][case]
    key, ns = info['key'], info['num_samples']
    g = OracleGraph(*golden_graph_parts(arrays, info['graph']))
    n_all = sum(len(v) for v in g.nodes.values())
    cursor = {v: 0 for v in range(n_all)}
    perm = cyclic.TapePerm(_split_perms(arrays, key))
    nested = []
    for k, roots in enumerate(info['calls']):
        walks = cyclic.walk_cyclic(g, ns, roots, cursor, perm)
        nested.append(walks)
        for h, hop in enumerate(walker.walks_to_hops(walks, ns)):
            np.testing.assert_array_equal(hop, arrays['%s_call%d_hop%d' % (key, k, h + 1)])
    assert perm.pos == info['n_perms']
    np.testing.assert_array_equal(np.asarray([cursor[v] for v in range(n_all)]), arrays[key + '_cursor'])
    if case == 0:
        assert nested == meta['cyclic0_nested']


def test_feistel_permutation_is_a_permutation():
    from oracle import cyclic
    for n in (1, 2, 3, 5, 16, 17, 100, 257):
        for t in (0, 1, 2 ** 33 + 5):
            img = [cyclic.feistel_index(t, n, i, 0x1234ABCD5678) for i in range(n)]
            assert sorted(img) == list(range(n))
    a = [cyclic.feistel_index(0, 100, i, 1) for i in range(100)]
    b = [cyclic.feistel_index(1, 100, i, 1) for i in range(100)]
    assert a != b and a != list(range(100))
