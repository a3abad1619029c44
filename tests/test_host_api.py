"""Host-side logic of graphsage_amd against the reference's recorded behaviour
(no GPU, no kernel calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_graph_parts
import graphsage_amd as ga
from graphsage_amd import _lib
from graphsage_amd.generator import GraphSAGEGenerator, LinkSequence
from oracle import generator as ogen
from oracle.graph import OracleGraph


def _err_graph():
    edges = np.array([[0, 1, 3, 3], [3, 3, 0, 1]])
    feats = {'user': np.zeros((3, 2)), 'item': np.zeros((2, 2))}
    return ga.Graph(feats, {'user': [0, 1, 2], 'item': [3, 4]}, edges)


def test_csr_matches_reference_neighbor_lists(golden):
    arrays, meta = golden
    feats = {'user': arrays['kat_feat_user'], 'item': arrays['kat_feat_item']}
    g = ga.Graph(feats, {'user': [0, 1, 2], 'item': [3, 4]}, arrays['kat_edges'])
    for node, neigh in meta['kat_neighbor_dict'].items():
        assert g.get_neighbors_from_node(int(node)) == neigh
    for name in ('g_small', 'g_mid'):
        parts = golden_graph_parts(arrays, name)
        g, o = ga.Graph(*parts), OracleGraph(*parts)
        assert g.indptr.dtype == np.int64 and g.indices.dtype == np.int32
        # the reference's own neighbor_dict, flattened when the fixtures were made
        np.testing.assert_array_equal(g.indptr, arrays[name + '_adj_indptr'])
        np.testing.assert_array_equal(g.indices, arrays[name + '_adj_indices'])
        for node, neigh in o.neighbor_dict.items():
            assert g.get_neighbors_from_node(node) == neigh
        assert g.num_nodes == o.num_nodes and g.num_edges == o.num_edges
        assert g.node_feature_sizes == o.node_feature_sizes
        ids = [g.num_users, g.num_users + 3]
        np.testing.assert_array_equal(g.get_node_features_from_node(ids, 'item'),
                                      o.get_node_features_from_node(ids, 'item'))
        np.testing.assert_array_equal(g.get_node_features_from_node([0, 7], 'user'),
                                      o.get_node_features_from_node([0, 7], 'user'))


def test_graph_surface_matches_recorded_reference(golden):
    _, meta = golden
    exp = meta['errors']
    g = _err_graph()
    assert repr(g) == exp['repr_graph']
    assert g.num_nodes == exp['num_nodes'] and g.num_edges == exp['num_edges']
    assert g.node_feature_sizes == exp['feature_sizes']
    assert [0 in g, 9 in g] == exp['contains']
    assert g.get_node_type(0) == 'user' and g.get_node_type(4) == 'item'
    with pytest.raises(KeyError):
        g.get_node_type(9)
    with pytest.raises(KeyError):                 # exp['no_out_edges'] == 'KeyError'
        g.get_neighbors_from_node(2)
    with pytest.raises(TypeError):
        ga.Graph([1, 2], {'user': [], 'item': []}, np.zeros((2, 0), dtype=int))


def test_isolated_node_error_is_both_reference_errors(golden):
    _, meta = golden
    assert meta['errors']['no_out_edges'] == 'KeyError'
    assert meta['errors']['empty_list'][0] == 'ValueError'
    err = ga.IsolatedNodeError("Samples must exist. Self-node also can be a sample (node 2 has no neighbors)")
    assert isinstance(err, KeyError) and isinstance(err, ValueError)
    assert str(err).startswith(meta['errors']['empty_list'][1])


def test_walker_and_generator_surface(golden):
    _, meta = golden
    exp = meta['errors']
    g = _err_graph()
    with pytest.raises(TypeError) as ei:
        ga.BreadthFirstWalker(g, (2, 2))
    assert [type(ei.value).__name__, str(ei.value)] == exp['walks_tuple']
    assert repr(ga.BreadthFirstWalker(g, [2, 2])) == exp['repr_walker']
    gen = GraphSAGEGenerator(g, 2, [2, 2], head_node_types=['user', 'item'])
    assert gen.multiplicity == exp['multiplicity'] and gen.num_batch_dims() == exp['num_batch_dims']
    assert [[t, l] for t, l in gen.type_adjacency_list] == meta['kat_type_adjacency_list']
    gen3 = GraphSAGEGenerator(g, 2, [3, 2], head_node_types=['user', 'item'])
    assert [[[t, list(l)] for t, l in row] for row in gen3._sampling_schema] == meta['kat_sampling_schema']
    # three hops: the same rule, checked against the oracle's statement of it
    gen4 = GraphSAGEGenerator(g, 2, [4, 3, 2])
    assert gen4.type_adjacency_list == ogen.type_adjacency_list(3)
    assert gen4._sampling_schema == ogen.sampling_schema([4, 3, 2])


def test_flow_validation_matches_reference(golden):
    _, meta = golden
    exp = meta['errors']
    g = _err_graph()
    gen = GraphSAGEGenerator(g, 2, [2, 2], head_node_types=['user', 'item'])
    for name, links in [('arity', [(0, 3, 1)]), ('unknown_src', [(9, 3)]), ('unknown_dst', [(0, 9)]),
                        ('wrong_types', [(3, 0)])]:
        want = exp['flow_' + name]
        with pytest.raises(Exception) as ei:
            gen.flow(links, targets=[1] * len(links))
        assert [type(ei.value).__name__, str(ei.value.args[0])] == want
    assert exp['flow_second_link_bad'] == 'ok:LinkSequence'
    assert isinstance(gen.flow([(0, 3), (3, 0)], targets=[1, 1]), LinkSequence)
    assert exp['flow_empty_returns'] == 'None' and gen.flow([], targets=[]) is None


def test_link_sequence_errors_and_order(golden):
    _, meta = golden
    exp = meta['errors']
    fn = lambda heads, bn: ['feats%d' % bn]
    for name, args in [('seq_len_mismatch', (fn, 2, [(0, 3)], [1, 0])),
                       ('seq_ids_not_iterable', (fn, 2, 5, None)),
                       ('seq_not_callable', (None, 2, [(0, 3)], [1]))]:
        with pytest.raises(Exception) as ei:
            LinkSequence(args[0], args[1], args[2], targets=args[3])
        assert [type(ei.value).__name__, str(ei.value)] == exp[name]
    seq = LinkSequence(fn, 2, [(0, 3)], targets=[1])
    with pytest.raises(IndexError) as ei:
        seq[1]
    assert [type(ei.value).__name__, str(ei.value)] == exp['seq_index']

    want = meta['sequence']
    calls = []

    def sample(heads, bn):
        calls.append([list(h) for h in heads])
        return ['feats%d' % bn]

    ids = [(i, 100 + i) for i in range(23)]
    targets = [i % 2 for i in range(23)]
    seq = LinkSequence(sample, 5, ids, targets=targets, shuffle=True, seed=0)
    assert len(seq) == want['len'] and seq.indices == want['epoch0']
    feats, t = seq[4]
    assert feats == ['feats4'] and [int(v) for v in t] == want['last_batch_targets']
    assert calls[-1] == want['last_batch_heads']
    seq.on_epoch_end()
    assert seq.indices == want['epoch1']
    assert LinkSequence(sample, 5, ids, targets=targets, shuffle=False, seed=0).indices == want['noshuffle']


def test_link_sequence_rank_sharding_partitions_each_epoch():
    ids = [(i, 100 + i) for i in range(23)]
    targets = list(range(23))
    fn = lambda heads, bn: None
    whole = LinkSequence(fn, 4, ids, targets=targets, shuffle=True, seed=5)
    parts = [LinkSequence(fn, 4, ids, targets=targets, shuffle=True, seed=5, rank=r, world_size=3) for r in range(3)]
    for _ in range(2):
        assert sorted(i for p in parts for i in p.indices) == list(range(23))
        for r, p in enumerate(parts):
            assert p.indices == whole.indices[r::3]
            assert len(p) == int(np.ceil(len(p.indices) / 4))
        whole.on_epoch_end()
        for p in parts:
            p.on_epoch_end()
    with pytest.raises(ValueError):
        LinkSequence(fn, 4, ids, targets=targets, shuffle=True, seed=None, rank=0, world_size=2)


def test_utils_surface():
    assert ga.is_real_iterable([1]) and not ga.is_real_iterable('ab') and not ga.is_real_iterable(3)
    import random
    assert ga.random_state(None).random is random
    a, b = ga.random_state(3), ga.random_state(3)
    assert a.random.random() == b.random.random()
    with pytest.raises(ValueError) as ei:
        ga.Aggregator(output_dim=31)
    assert str(ei.value) == "The output_dim must be a multiple of two."
    with pytest.raises(AssertionError):
        ga.Aggregator(agg_option='sum')


def test_model_plan_matches_oracle_restatement_of_reference_wiring():
    from oracle.model import LayerPlan
    g = _err_graph()
    for ns, sizes in [([8, 4], [32, 32]), ([25, 10, 5], [16, 8, 4])]:
        gen = GraphSAGEGenerator(g, 2, ns)
        m = ga.GraphSAGE(sizes, gen)
        plan = LayerPlan(sizes, ns, g.node_feature_sizes)
        assert m.neigh_trees == plan.neigh_trees
        assert m._depths == plan.depths
        assert m.dims == plan.dims
        assert m.activations == plan.activations
        assert m._input_shapes() == [(s, g.node_feature_sizes[t]) for s, t in
                                     zip([1, 1] + [v for p in np.cumprod(ns) for v in (int(p), int(p))],
                                         [t for t, _ in gen.type_adjacency_list])]
    with pytest.raises(ValueError):
        ga.GraphSAGE([32, 32], None)


def test_library_exports_every_declared_symbol():
    """The C ABI library loads on a machine without a GPU and exports exactly the
    entry points include/gsage.h declares (nothing is called)."""
    header = open(os.path.join(ROOT, 'include', 'gsage.h')).read()
    declared = set(re.findall(r'\b(gs_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert os.path.exists(_lib.LIB_PATH), "build first: make -C graphsage_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.gs_version.restype = ctypes.c_int
    assert lib.gs_version() >= 1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = _err_graph()
    with pytest.raises(_lib.GsageError):
        ga.BreadthFirstWalker(g, [2, 2]).run_breadth_first_walk([0])


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'graphsage_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+\.*oracle', text, re.M), f
