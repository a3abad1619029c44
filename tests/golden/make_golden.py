#!/usr/bin/env python3
"""Regenerates the fixtures in this directory by running the reference itself.

Run in the build container only (it needs the reference checkout; nothing on
the GPU box does):

    python tests/golden/make_golden.py [/root/reference]

What is imported: the reference's ``bipartite_graph.py``, ``sampler.py``,
``generator.py`` and ``backup/sampler_backup.py``, unmodified.  ``generator.py`` and ``utils.py`` import
TensorFlow for two base classes (``Sequence``, ``Layer``) that play no part in
batching or sampling; TensorFlow is not installed here, so empty stand-in
classes are registered under those module names before the import.  No
TensorFlow arithmetic is run or faked -- the model side has no fixture.

Fixtures are data only: inputs (edges, features, links, tapes of pre-drawn
32-bit words) and the outputs the reference produced for them.
"""
import json
import os
import random
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_tensorflow():
    tf = types.ModuleType('tensorflow')
    keras = types.ModuleType('tensorflow.keras')
    k_utils = types.ModuleType('tensorflow.keras.utils')
    k_layers = types.ModuleType('tensorflow.keras.layers')
    k_backend = types.ModuleType('tensorflow.keras.backend')
    k_act = types.ModuleType('tensorflow.keras.activations')

    class Sequence:
        pass

    class Layer:
        def __init__(self, **kwargs):
            pass

    k_utils.Sequence = Sequence
    k_layers.Layer = Layer
    keras.utils, keras.layers, keras.backend, keras.activations = k_utils, k_layers, k_backend, k_act
    tf.keras = keras
    for name, mod in [('tensorflow', tf), ('tensorflow.keras', keras),
                      ('tensorflow.keras.utils', k_utils), ('tensorflow.keras.layers', k_layers),
                      ('tensorflow.keras.backend', k_backend),
                      ('tensorflow.keras.activations', k_act)]:
        sys.modules[name] = mod


_stub_tensorflow()
sys.path.insert(0, REF)
import bipartite_graph as ref_graph      # noqa: E402
import sampler as ref_sampler            # noqa: E402
import generator as ref_generator        # noqa: E402


class TapeRS:
    """``choices`` fed from pre-drawn 32-bit words: pop[(r * len(pop)) >> 32]."""

    def __init__(self, tape):
        self.tape = [int(t) for t in tape]
        self.pos = 0

    def choices(self, population, k=1):
        n = len(population)
        out = []
        for _ in range(k):
            out.append(population[(self.tape[self.pos] * n) >> 32])
            self.pos += 1
        return out


def hop_sizes(ns):
    out, p = [], 1
    for n in ns:
        p *= n
        out.append(p)
    return out


def flatten_walks(walks, ns):
    sizes = hop_sizes(ns)
    hops = [np.zeros((len(walks), s), dtype=np.int64) for s in sizes]
    for w, walk in enumerate(walks):
        pos, groups = 1, 1
        for h in range(len(ns)):
            flat = []
            for g in range(groups):
                flat.extend(walk[pos + g])
            hops[h][w] = flat
            pos += groups
            groups = sizes[h]
    return hops


def make_graph(rng, nu, ni, n_edges, fu, fi, hub=True, dup=True):
    # random edges among all but the last user and the last item
    u = rng.integers(0, nu - 1, n_edges)
    it = rng.integers(0, ni - 1, n_edges) + nu
    if hub:                                   # first item connected to every such user
        u = np.concatenate([u, np.arange(nu - 1)])
        it = np.concatenate([it, np.full(nu - 1, nu)])
    # every node gets at least one edge ...
    u = np.concatenate([u, np.arange(nu - 1), rng.integers(0, nu - 1, ni - 1)])
    it = np.concatenate([it, nu + (np.arange(nu - 1) % (ni - 1)), nu + np.arange(ni - 1)])
    # ... and the last user and the last item exactly one (each other): degree 1
    u = np.concatenate([u, [nu - 1]])
    it = np.concatenate([it, [nu + ni - 1]])
    if dup:                                   # repeated edges raise sampling weight
        u = np.concatenate([u, u[:17]])
        it = np.concatenate([it, it[:17]])
    perm = rng.permutation(len(u))
    e = np.stack([u[perm], it[perm]])
    edges = np.concatenate([e, np.flip(e, axis=0)], axis=1).astype(np.int64)
    feats = {'user': rng.standard_normal((nu, fu)).astype(np.float32),
             'item': rng.standard_normal((ni, fi)).astype(np.float32)}
    nodes = {'user': list(range(nu)), 'item': list(range(nu, nu + ni))}
    return feats, nodes, edges


def kat_cases(meta):
    e = np.array([[0, 0, 1, 2, 2], [3, 4, 3, 4, 3]])
    edges = np.concatenate([e, np.flip(e, 0)], axis=1)
    feats = {'user': np.arange(6, dtype=np.float64).reshape(3, 2),
             'item': (np.arange(6, dtype=np.float64) + 100).reshape(2, 3)}
    nodes = {'user': [0, 1, 2], 'item': [3, 4]}
    g = ref_graph.Graph(feats, nodes, edges)
    meta['kat_neighbor_dict'] = {str(k): v for k, v in g.neighbor_dict.items()}

    random.seed(7)
    w = ref_sampler.BreadthFirstWalker(g, [3, 2])
    meta['kat_walk_seed7'] = w.run_breadth_first_walk([0, 4])

    # the uniform stream the walk above consumed, to pin choices() == floor(u*n)
    random.seed(7)
    meta['kat_uniforms_seed7'] = [random.random() for _ in range(2 * (3 + 6))]

    random.seed(3)
    gen = ref_generator.GraphSAGEGenerator(g, 2, [3, 2], head_node_types=['user', 'item'])
    seq = gen.flow([(0, 3), (2, 4), (1, 3)], targets=[1, 0, 1])
    feats0, targets0 = seq[0]
    meta['kat_gen_seed3_shapes'] = [list(a.shape) for a in feats0]
    meta['kat_gen_seed3_targets'] = [int(t) for t in targets0]
    meta['kat_gen_seed3_len'] = len(seq)
    meta['kat_type_adjacency_list'] = [[t, l] for t, l in gen.type_adjacency_list]
    meta['kat_sampling_schema'] = [[[t, list(l)] for t, l in row] for row in gen._sampling_schema]
    arrays = {'kat_gen_seed3_x%d' % i: a for i, a in enumerate(feats0)}
    arrays['kat_edges'] = edges
    arrays['kat_feat_user'] = feats['user']
    arrays['kat_feat_item'] = feats['item']
    return arrays


def replay_cases(meta):
    arrays = {}
    rng = np.random.default_rng(20240611)
    graphs = {
        'g_small': make_graph(rng, 40, 12, 150, 6, 5),
        'g_mid': make_graph(rng, 700, 90, 6000, 8, 4),
    }
    cases = [
        ('g_small', [3, 2], 1), ('g_small', [3, 2], 7), ('g_small', [8, 4], 7),
        ('g_mid', [8, 4], 100), ('g_mid', [10, 5], 100), ('g_mid', [25, 10], 7),
        ('g_mid', [25, 10], 1),
    ]
    walker_only = [('g_small', [4, 3, 2], 7), ('g_mid', [4, 3, 2], 33), ('g_mid', [70, 3], 5)]
    built = {}
    for name, (feats, nodes, edges) in graphs.items():
        built[name] = ref_graph.Graph(feats, nodes, edges)
        arrays[name + '_edges'] = edges
        arrays[name + '_feat_user'] = feats['user']
        arrays[name + '_feat_item'] = feats['item']
        # the reference's adjacency itself: neighbor_dict flattened to (row starts, neighbors)
        nd = built[name].neighbor_dict
        n_all = feats['user'].shape[0] + feats['item'].shape[0]
        lists = [list(nd.get(v, [])) for v in range(n_all)]
        arrays[name + '_adj_indptr'] = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int64)
        arrays[name + '_adj_indices'] = np.asarray([x for l in lists for x in l], dtype=np.int32)
    meta['replay'] = []
    for ci, (gname, ns, b) in enumerate(cases):
        g = built[gname]
        nu = g.node_features['user'].shape[0]
        e = arrays[gname + '_edges']
        pick = rng.integers(0, e.shape[1] // 2, b)
        links = [(int(e[0, p]), int(e[1, p])) for p in pick]      # first half = user->item
        if b >= 7:                             # the degree-1 pair and the hub item as heads
            ni = g.node_features['item'].shape[0]
            links[0] = (nu - 1, nu + ni - 1)
            links[1] = (links[1][0], nu)
        t = sum(hop_sizes(ns))
        tape = rng.integers(0, 2 ** 32, 2 * b * t, dtype=np.uint64).astype(np.uint32)
        if ci % 2 == 0:                        # extreme words: first/last neighbor
            tape[:5] = [0, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 1]
        gen = ref_generator.GraphSAGEGenerator(g, b, ns, head_node_types=['user', 'item'])
        rs = TapeRS(tape)
        gen.sampler.rs = rs
        feats_out = gen.sample_features(links, 0)
        assert rs.pos == len(tape)
        # the walks themselves, for the sampler-level check
        walker = ref_sampler.BreadthFirstWalker(g, ns)
        walker.rs = TapeRS(tape)
        roots = [l[0] for l in links] + [l[1] for l in links]
        walks = walker.run_breadth_first_walk(roots)
        hops = flatten_walks(walks, ns)
        key = 'replay%d' % ci
        meta['replay'].append({'key': key, 'graph': gname, 'num_samples': ns, 'batch': b,
                               'num_users': nu})
        arrays[key + '_links'] = np.asarray(links, dtype=np.int64)
        arrays[key + '_tape'] = tape
        for h, hop in enumerate(hops):
            arrays[key + '_hop%d' % (h + 1)] = hop.astype(np.int32)
        for i, a in enumerate(feats_out):
            arrays[key + '_x%d' % i] = a
    meta['walker_only'] = []
    for ci, (gname, ns, nroots) in enumerate(walker_only):
        g = built[gname]
        roots = [int(r) for r in rng.integers(0, g.num_nodes, nroots)]
        t = sum(hop_sizes(ns))
        tape = rng.integers(0, 2 ** 32, nroots * t, dtype=np.uint64).astype(np.uint32)
        walker = ref_sampler.BreadthFirstWalker(g, ns)
        walker.rs = TapeRS(tape)
        walks = walker.run_breadth_first_walk(roots)
        hops = flatten_walks(walks, ns)
        key = 'walk%d' % ci
        meta['walker_only'].append({'key': key, 'graph': gname, 'num_samples': ns})
        arrays[key + '_roots'] = np.asarray(roots, dtype=np.int64)
        arrays[key + '_tape'] = tape
        for h, hop in enumerate(hops):
            arrays[key + '_hop%d' % (h + 1)] = hop.astype(np.int32)
        if ci == 0:
            meta['walk0_nested'] = walks
    return arrays


def error_cases(meta):
    out = {}
    # (a) node with no outgoing edge, graph built by the constructor
    edges = np.array([[0, 1, 3, 3], [3, 3, 0, 1]])      # item 4 and user 2 never a source
    feats = {'user': np.zeros((3, 2)), 'item': np.zeros((2, 2))}
    nodes = {'user': [0, 1, 2], 'item': [3, 4]}
    g = ref_graph.Graph(feats, nodes, edges)
    w = ref_sampler.BreadthFirstWalker(g, [2, 2])
    try:
        w.run_breadth_first_walk([2])
        out['no_out_edges'] = 'no error'
    except Exception as ex:                             # noqa: BLE001
        out['no_out_edges'] = type(ex).__name__
    # (b) explicit empty neighbor list
    g.neighbor_dict[2] = []
    try:
        w.run_breadth_first_walk([2])
        out['empty_list'] = 'no error'
    except Exception as ex:                             # noqa: BLE001
        out['empty_list'] = [type(ex).__name__, str(ex)]
    # (c) num_of_walks not a list
    try:
        ref_sampler.BreadthFirstWalker(g, (2, 2))
        out['walks_tuple'] = 'no error'
    except Exception as ex:                             # noqa: BLE001
        out['walks_tuple'] = [type(ex).__name__, str(ex)]
    # (d) flow() validation
    gen = ref_generator.GraphSAGEGenerator(g, 2, [2, 2], head_node_types=['user', 'item'])
    for name, links in [('arity', [(0, 3, 1)]), ('unknown_src', [(9, 3)]), ('unknown_dst', [(0, 9)]),
                        ('wrong_types', [(3, 0)]), ('second_link_bad', [(0, 3), (3, 0)])]:
        try:
            r = gen.flow(links, targets=[1] * len(links))
            out['flow_' + name] = 'ok:' + type(r).__name__
        except Exception as ex:                         # noqa: BLE001
            out['flow_' + name] = [type(ex).__name__, str(ex.args[0])]
    out['flow_empty_returns'] = repr(gen.flow([], targets=[]))
    try:
        ref_generator.LinkSequence(gen.sample_features, 2, [(0, 3)], targets=[1, 0])
    except Exception as ex:                             # noqa: BLE001
        out['seq_len_mismatch'] = [type(ex).__name__, str(ex)]
    try:
        ref_generator.LinkSequence(gen.sample_features, 2, 5, targets=None)
    except Exception as ex:                             # noqa: BLE001
        out['seq_ids_not_iterable'] = [type(ex).__name__, str(ex)]
    try:
        ref_generator.LinkSequence(None, 2, [(0, 3)], targets=[1])
    except Exception as ex:                             # noqa: BLE001
        out['seq_not_callable'] = [type(ex).__name__, str(ex)]
    seq = ref_generator.LinkSequence(gen.sample_features, 2, [(0, 3)], targets=[1])
    try:
        seq[1]
    except Exception as ex:                             # noqa: BLE001
        out['seq_index'] = [type(ex).__name__, str(ex)]
    out['repr_graph'] = repr(g)
    out['repr_walker'] = repr(w)
    out['num_nodes'] = g.num_nodes
    out['num_edges'] = g.num_edges
    out['feature_sizes'] = g.node_feature_sizes
    out['contains'] = [0 in g, 9 in g]
    out['multiplicity'] = gen.multiplicity
    out['num_batch_dims'] = gen.num_batch_dims()
    meta['errors'] = out


def sequence_cases(meta):
    calls = []

    def sample(head_ids, batch_num):
        calls.append(([list(h) for h in head_ids], batch_num))
        return ['feats%d' % batch_num]

    ids = [(i, 100 + i) for i in range(23)]
    targets = [i % 2 for i in range(23)]
    out = {}
    seq = ref_generator.LinkSequence(sample, 5, ids, targets=targets, shuffle=True, seed=0)
    out['len'] = len(seq)
    out['epoch0'] = list(seq.indices)
    b4 = seq[4]
    out['last_batch_targets'] = [int(t) for t in b4[1]]
    out['last_batch_heads'] = calls[-1][0]
    seq.on_epoch_end()
    out['epoch1'] = list(seq.indices)
    seq2 = ref_generator.LinkSequence(sample, 5, ids, targets=targets, shuffle=False, seed=0)
    out['noshuffle'] = list(seq2.indices)
    meta['sequence'] = out


class TapePermutation:
    """Stands in for the ``numpy.random`` module inside the reference's backup/sampler_backup.py
    (its ``np_rn``): ``permutation(x)`` applies the next pre-drawn index permutation."""

    def __init__(self, perms):
        self.perms = perms
        self.pos = 0

    def permutation(self, x):
        p = self.perms[self.pos]
        self.pos += 1
        assert len(p) == len(x)
        return np.asarray(x)[p]


def cyclic_cases(meta):
    """The reference's alternate walker (backup/sampler_backup.py), unmodified, on its own Graph (whose
    sampling_index_dict it advances), with numpy's permutations replaced by pre-drawn ones: a first pass
    draws and records them, a second pass on a fresh graph replays the record and must give the same walks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_sampler_backup', os.path.join(REF, 'backup', 'sampler_backup.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(77001)
    arrays = {}
    meta['cyclic'] = []
    graphs = {'g_small': make_graph(np.random.default_rng(20240611), 40, 12, 150, 6, 5)}
    cases = [('g_small', [3, 2], [[0, 51, 7], [7, 7, 39]]),            # two calls in a row: the cursors carry over
             ('g_small', [8, 4], [[39, 51, 3, 3]]),                     # user 39 / item 51: one neighbor each, k > length
             ('g_small', [4, 3, 2], [[5, 45], [45, 5, 40]]),
             ('g_small', [70], [[40, 2]])]                               # more samples than lanes in a wavefront
    for ci, (gname, ns, calls) in enumerate(cases):
        feats, nodes, edges = graphs[gname]
        # pass 1 draws the permutations (and records them); pass 2 replays them on a fresh graph
        drawn = []

        class Drawer:
            def permutation(self, x):
                p = rng.permutation(len(x))
                drawn.append(p)
                return np.asarray(x)[p]
        g = ref_graph.Graph(feats, nodes, edges)
        mod.np_rn = Drawer()
        w = mod.BreadthFirstWalker(g, ns)
        first = [w.run_breadth_first_walk(list(c)) for c in calls]
        g2 = ref_graph.Graph(feats, nodes, edges)
        mod.np_rn = TapePermutation(drawn)
        w2 = mod.BreadthFirstWalker(g2, ns)
        again = [w2.run_breadth_first_walk(list(c)) for c in calls]
        assert again == first and mod.np_rn.pos == len(drawn)
        key = 'cyclic%d' % ci
        meta['cyclic'].append({'key': key, 'graph': gname, 'num_samples': ns, 'calls': [list(map(int, c)) for c in calls],
                               'n_perms': len(drawn)})
        arrays[key + '_perm_lengths'] = np.asarray([len(p) for p in drawn], dtype=np.int64)
        arrays[key + '_perms'] = (np.concatenate(drawn) if drawn else np.zeros(0)).astype(np.int32)
        for k, walks in enumerate(first):
            for h, hop in enumerate(flatten_walks(walks, ns)):
                arrays['%s_call%d_hop%d' % (key, k, h + 1)] = hop.astype(np.int32)
        n_all = feats['user'].shape[0] + feats['item'].shape[0]
        arrays[key + '_cursor'] = np.asarray([g.sampling_index_dict[v] for v in range(n_all)], dtype=np.int64)
        if ci == 0:
            meta['cyclic0_nested'] = first
    return arrays


def main():
    meta = {}
    arrays = {}
    arrays.update(kat_cases(meta))
    arrays.update(replay_cases(meta))
    error_cases(meta)
    sequence_cases(meta)
    arrays.update(cyclic_cases(meta))
    np.savez_compressed(os.path.join(OUT, 'reference_vectors.npz'), **arrays)
    with open(os.path.join(OUT, 'reference_vectors.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    size = os.path.getsize(os.path.join(OUT, 'reference_vectors.npz'))
    print('wrote %d arrays, %.1f KiB' % (len(arrays), size / 1024))


if __name__ == '__main__':
    main()
