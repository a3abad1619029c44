"""Static launch plan for one training step of the GraphSAGE stack.

Forward: for two hops, the one-kernel walk forward, which also writes what the
backward pass reads; for other depths the sampler kernel, then one fused kernel
per (layer, node type) that saves the same things (layer 0: the gathered own
rows and neighbor means, so nothing is gathered twice; layers >= 1: the
neighbor means).
Backward: per (layer, node type) the MFMA weight-gradient kernel with its
ordered slab reduction, the input-gradient GEMM for layers >= 1, and the
broadcast that undoes the dense neighbor mean.  Weight gradients are written
straight into the views of the flat all-reduce bucket.  Feature tables are
inputs of the reference model, not variables (``models.py:241``), so nothing is
scattered back into them.
"""
import torch

from . import kernels
from .engine import SageForwardEngine, _TYPES


class SageTrainEngine(SageForwardEngine):
    def __init__(self, graph, model, batch_size):
        # Feature widths that are not multiples of 32 (MovieLens: 28 / 18 columns) run through zero-padded
        # copies of the tables and of the layer-0 weights; the real weights and their gradients are copied
        # in and out around every pass (padding.py).
        self.dropout = float(getattr(model, 'dropout', 0.0) or 0.0)
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError("dropout rate must lie in [0, 1)")
        self._padded_model = None
        f = graph.node_feature_sizes
        if any(f[t] % 32 for t in _TYPES):
            from .padding import PaddedGraphView, PaddedModelView, padded_widths
            wide = padded_widths(f)
            if all(kernels.fused_supported(wide[t], wide['item' if t == 'user' else 'user'], model.build().dims[1][t],
                                           max(model.n_samples)) for t in _TYPES):
                graph = PaddedGraphView(graph, wide)
                model = self._padded_model = PaddedModelView(model, wide)
        super().__init__(graph, model, batch_size, use_fused=True, walk_kernel=False)
        if self.agg_max:
            raise kernels._lib.GsageError("the training plan covers the mean aggregator; agg_option='max' trains "
                                          "through graphsage_amd.train.Trainer(engine='autograd')")
        f = graph.node_feature_sizes
        # two hops: sampling, both layers and everything the backward pass reads in ONE kernel
        self.walk_train = bool(self.n_hops == 2 and kernels.sage2_supported(
            f['user'], f['item'], self.model.dims[1]['user'], self.model.dims[2]['user'], *self.fanouts))
        if not all(self.fused_layers) and not self.walk_train:
            raise kernels._lib.GsageError(
                "the training plan needs a two-hop shape of the one-kernel forward or a fused layer kernel for every "
                "layer (widths 32/64/128 with half-widths 16 or 32, or multiples of 32 with half-widths that are "
                "multiples of 128); use "
                "graphsage_amd.train.Trainer(engine='autograd') for other shapes")
        m = self.model
        dev = self.indptr.device
        new = lambda r, c: torch.empty((r, c), dtype=torch.float32, device=dev)
        self.z = {t: new(self.batch, m.dims[-1][t]) for t in _TYPES}          # last layer before l2norm
        self.g_z = {t: new(self.batch, m.dims[-1][t]) for t in _TYPES}
        self.g_h = [{t: torch.zeros_like(self.h[layer][t]) for t in _TYPES} for layer in range(self.n_hops)]
        self.g_xs = [None] + [{t: new(self.h[layer][t].shape[0], m.dims[layer][t]) for t in _TYPES}
                              for layer in range(1, self.n_hops)]
        self.g_xm = [None] + [{t: new(self.h[layer][t].shape[0], m.dims[layer][self.other[t]]) for t in _TYPES}
                              for layer in range(1, self.n_hops)]
        self.ws = [{t: torch.empty(kernels.project_bwd_workspace_floats(
            m.dims[layer][t], m.dims[layer][self.other[t]], m.dims[layer + 1][t], self.h[layer][t].shape[0]),
            dtype=torch.float32, device=dev) for t in _TYPES} for layer in range(self.n_hops)]
        self._train_calls = {}
        self._side = None                                       # second stream of the backward pass
        self._walk_train_call = None
        # ... and then the top of the backward pass in one kernel as well (see _backward_walks)
        self._tail_kernel = self.walk_train and kernels.sage2_backward_tail_supported(m.dims[1]['user'], m.dims[2]['user'])
        self._side2 = None
        self._wgrad_pairs = None
        # ... and, where a wavefront can keep every parameter's sum in registers, the WHOLE backward pass in one
        # kernel and one reduction (see _backward_walks); fused_backward = False keeps the separate launches
        self.fused_backward = bool(self._tail_kernel and kernels.sage2_backward_all_supported(
            f['user'], f['item'], m.dims[1]['user'], m.dims[2]['user']))
        self._bwd_all = None
        # Own rows of layer 0 are not written by the forward pass when the weight-gradient kernel can read
        # them from the feature table by node id: the forward kernel then leaves 4 bytes per row (the id)
        # instead of the row.
        self.self_ids = None
        if (self._tail_kernel and f['user'] == f['item'] and
                kernels.load().gs_sage_project_wgrad_by_id_supported(f['user'], f['item'], m.dims[1]['user'])):
            self.self_ids = {t: torch.zeros(self.h[0][t].shape[0], dtype=torch.int32, device=dev) for t in _TYPES}
        # The hop-1 heads' layer-0 outputs are read back only for their relu mask (their values reach layer 1
        # through the saved neighbor mean): the forward kernel leaves one bit per element instead of the rows,
        # the top-of-backward kernel applies the masks, and the layer-0 weight gradients take its output as it is.
        self.h1_mask = None
        if self._tail_kernel and all(m._aggs[0][t].activation == 'relu' for t in _TYPES):
            words = self.batch * self.fanouts[0] * (m.dims[1]['user'] // 32)
            self.h1_mask = {t: torch.zeros(words, dtype=torch.int32, device=dev) for t in _TYPES}
        self._head_slab = None
        # The layered routes (other depths) do the same for layer 0 where its weight-gradient kernel reads own rows
        # by id: the node ids of all layer-0 heads are put together in row order once per step (a few small copies),
        # the layer's forward kernel writes no copy of the own rows, and both node types' weight gradients are one launch.
        self.row_ids = None
        self._wgrad_l0_pair = None
        if (not self.walk_train and f['user'] == f['item'] and self.fused_layers[0] and not self.wide_layers[0] and
                all(m._aggs[0][t].activation == 'relu' for t in _TYPES) and
                kernels.load().gs_sage_project_wgrad_by_id_supported(f['user'], f['item'], m.dims[1]['user'])):
            self.row_ids = {t: torch.zeros(self.h[0][t].shape[0], dtype=torch.int32, device=dev) for t in _TYPES}
        # Three hops with 32 units per layer: the whole backward pass on these tables is two kernels and a reduction
        # (gs_sage3_backward_all); fused_backward3 = False keeps the layer-by-layer launches
        self.fused_backward3 = bool(
            self.row_ids is not None and self.n_hops == 3 and
            all(m.dims[layer][t] == 32 for layer in (1, 2, 3) for t in _TYPES) and
            all(m._aggs[layer][t].activation == 'relu' for layer in (0, 1) for t in _TYPES) and
            kernels.load().gs_sage3_backward_all_supported(f['user'], f['item'], 32, 32, 32, self.fanouts[0], self.fanouts[1]))
        self._bwd3 = None
        # ... and the forward pass below the seeds one kernel as well (the subtree kernel in its saving form: hops 2
        # and 3 drawn where they are used, the hop-2 nodes' outputs kept as one mask word each), then one kernel over
        # the seeds; subtree_train = False keeps the layer-by-layer forward
        self.subtree_train = bool(
            self.fused_backward3 and self._padded_model is None and not self.dropout and
            kernels.sage2_subtrees_supported(f['user'], f['item'], 32, 32, self.fanouts[1], self.fanouts[2]) and
            kernels.sage3_seeds_supported(f['user'], f['item'], 32, 32, 32, self.fanouts[0]))
        self._sub_train = None
        # Dropout of the aggregator inputs (models.py:152,165) is drawn inside the one-kernel forward; what that
        # kernel saves are the inputs as they entered the layers, so the own rows are saved as rows (not ids) and
        # the masks carry "active and kept".
        if self.dropout:
            if not (self.walk_train and self._tail_kernel and self.h1_mask is not None):
                raise kernels._lib.GsageError(
                    "the training plan draws dropout in its two-hop one-kernel route only; this model trains through "
                    "graphsage_amd.train.Trainer(engine='autograd')")
            self.self_ids = None

    @property
    def tail_kernel(self):
        return self._tail_kernel

    @tail_kernel.setter
    def tail_kernel(self, on):
        """Switching the one-launch top of the backward pass off (comparisons, tests) also brings back
        the gathered copy of the own rows, which the layer-by-layer backward kernels read."""
        if self.dropout and not on:
            raise kernels._lib.GsageError("dropout is drawn by the one-kernel route only")
        self._tail_kernel = bool(on) and self.walk_train
        if not self._tail_kernel and (self.self_ids is not None or self.h1_mask is not None):
            self.self_ids = None
            self.h1_mask = None
            self._walk_train_call = None
            self._wgrad_pairs = None
            self._bwd_all = None

    def _forward_walks_train(self, seeds, walk_ids, rng_seed, replay):
        m = self.model
        if self._walk_train_call is None:
            weights = {(l, t): (m._aggs[l][t].w_self, m._aggs[l][t].w_neigh, m._aggs[l][t].bias)
                       for l in (0, 1) for t in _TYPES}
            rows = {t: self.h[0][t].shape[0] for t in _TYPES}
            saved = {'x_self': self.x_self, 'x_mean': {t: self.x_mean[0][t][:rows[t]] for t in _TYPES},
                     'h1': self.h[0], 'x_mean1': {t: self.x_mean[1][t][:self.batch] for t in _TYPES}, 'z': self.z}
            if self.self_ids is not None:
                saved['self_ids'] = self.self_ids
                del saved['x_self']
            if self.h1_mask is not None:
                saved['h1_mask'] = self.h1_mask
            if self.dropout:
                saved['dropout'] = self.dropout
            self._walk_train_call = kernels.Sage2Call(
                self.indptr, self.indices, self.tables['user'][0], self.tables['item'][0], self.fanouts, weights,
                m.normalize == "l2", self.batch, emb=self.emb_all, saved=saved)
        self._walk_train_call.run(seeds, self.err_flag, rng_seed=rng_seed, walk_ids=walk_ids, replay=replay)
        return self.emb_all[:self.batch], self.emb_all[self.batch:]

    # ---- forward ------------------------------------------------------------------
    def _train_call(self, layer, t):
        key = (layer, t)
        if key in self._train_calls:
            return self._train_calls[key]
        m, other = self.model, self.other
        agg = m._aggs[layer][t]
        last = layer == self.n_hops - 1
        rows = self.h[layer][t].shape[0]
        segs = self._layer_segments(layer, t)
        if layer == 0:
            (f_self, o_self), (f_neigh, o_neigh) = self.tables[t], self.tables[other[t]]
            save_self = None if self.row_ids is not None else self.x_self[t]
        else:
            f_self, o_self, f_neigh, o_neigh = self.h[layer - 1][t], 0, self.h[layer - 1][other[t]], 0
            save_self = None                                   # the own rows are h[layer-1] itself
        out = self.z[t] if last else self.h[layer][t]
        call = kernels.FusedLayerCall(f_self, o_self, f_neigh, o_neigh, segs, agg.w_self, agg.w_neigh, agg.bias,
                                      agg.activation, False, out, save_self=save_self,
                                      save_mean=self.x_mean[layer][t][:rows],
                                      sampling=self._sampling_desc() if layer == 0 else None)
        self._train_calls[key] = call
        return call

    def forward_train(self, seeds, walk_ids=None, rng_seed=0, replay=None):
        """Returns the (user, item) embeddings; every intermediate the backward
        pass needs stays in the engine's buffers."""
        if self._padded_model is not None:
            self._padded_model.refresh()
        if self.walk_train:
            return self._forward_walks_train(seeds, walk_ids, rng_seed, replay)
        if self.subtree_train and self.fused_backward3:
            return self._forward_subtrees_train(seeds, walk_ids, rng_seed, replay)
        self.sample(seeds, walk_ids, rng_seed, replay)
        if self.row_ids is not None:                         # the layer-0 heads' node ids in row order, for the backward pass
            for t in _TYPES:
                parts = [(r0, self._slot_ids(seeds, h, side)) for (_, h, side, r0, n) in self.segments[t] if h <= self.n_hops - 1]
                torch.cat([ids for _, ids in sorted(parts, key=lambda x: x[0])], out=self.row_ids[t])    # one launch per type
        for layer in range(self.n_hops):
            if layer > 0 and self.wide_first_layer_only:
                # wide models: the later layers have few rows; their neighbor means are needed in memory for
                # the backward pass anyway, so they take gather-mean + the tiled projection
                self._general_layer_train(layer)
                continue
            for t in _TYPES:
                call = self._train_call(layer, t)
                if layer == 0 and seeds.data_ptr() != self.seeds_buf.data_ptr():
                    for i, (_, h, side, _, _) in enumerate(self.segments[t]):
                        if h == 0:
                            call.set_head_ids(i, self._slot_ids(seeds, 0, side))
                if layer == 0 and self.draw_deepest:
                    call.run(rng_seed=rng_seed, walk_ids=walk_ids, replay=replay)
                else:
                    call.run()
        for t in _TYPES:
            if self.model.normalize == "l2":
                kernels.l2_normalize(self.z[t], out=self.emb[t])
            else:
                self.emb[t].copy_(self.z[t])
        return self.emb['user'], self.emb['item']

    def _forward_subtrees_train(self, seeds, walk_ids, rng_seed, replay):
        """Three hops, two launches: everything below the seeds (all three hops drawn; layers 0 and 1 per hop-1 node's
        subtree), then the three layers at the seeds; both leave what ``_backward_three`` reads in the engine's tables
        (the rows of the layer-by-layer route, except that the hop-1 nodes' outputs are two arrays over both node
        types and the hop-2 nodes' are mask words)."""
        m, b = self.model, self.batch
        n0, n1, n2 = self.fanouts
        if self._sub_train is None:
            dev = self.indptr.device
            r1, r2 = b * (1 + n0), b * (1 + n0 + n0 * n1)
            st = {'h0': torch.empty((2 * b * n0, 32), dtype=torch.float32, device=dev),
                  'h1': torch.empty((2 * b * n0, 32), dtype=torch.float32, device=dev),
                  'mask': {t: torch.zeros(b * n0 * n1, dtype=torch.int32, device=dev) for t in _TYPES},
                  'hop1': self.hops[0].reshape(-1), 'hop2': self.hops[1].reshape(-1)}
            weights = {(l, t): (m._aggs[l][t].w_self, m._aggs[l][t].w_neigh, m._aggs[l][t].bias)
                       for l in (0, 1, 2) for t in _TYPES}
            below = {'x_mean0_hop2': {t: self.x_mean[0][t][r1:r2] for t in _TYPES}, 'mask_hop2': st['mask'],
                     'x_mean0_hop1': {t: self.x_mean[0][t][b:r1] for t in _TYPES},
                     'x_mean1_hop1': {t: self.x_mean[1][t][b:r1] for t in _TYPES}}
            st['call'] = kernels.Sage2Call(
                self.indptr, self.indices, self.tables['user'][0], self.tables['item'][0], [n1, n2],
                {k: w for k, w in weights.items() if k[0] < 2}, False, b, emb=st['h1'], hop_buffers=(st['hop2'], None),
                subtrees={'roots_per_walk': n0, 'tape_len': sum(self.sizes[1:]), 'tape_off_hop2': self.sizes[1],
                          'tape_off_hop3': self.sizes[1] + self.sizes[2], 'out_h0': st['h0'], 'draw_hop1': st['hop1'],
                          'saved': below})
            at_seeds = {'x_mean0': {t: self.x_mean[0][t][:b] for t in _TYPES}, 'h0': {t: self.h[0][t][:b] for t in _TYPES},
                        'x_mean1': {t: self.x_mean[1][t][:b] for t in _TYPES}, 'h1': {t: self.h[1][t][:b] for t in _TYPES},
                        'x_mean2': {t: self.x_mean[2][t][:b] for t in _TYPES}, 'z': self.z}
            st['seeds'] = kernels.Sage3SeedsCall(
                self.tables['user'][0], self.tables['item'][0], st['hop1'], n0, st['h0'], st['h1'], weights,
                m._aggs[2]['user'].activation, m.normalize == "l2", b, self.emb_all, saved=at_seeds)
            self._sub_train = st
        st = self._sub_train
        if seeds.data_ptr() != self.seeds_buf.data_ptr():     # (the backward pass reads the seeds' own rows by these ids)
            self.seeds_buf.copy_(seeds)
        st['call'].run(self.seeds_buf, self.err_flag, rng_seed=rng_seed, walk_ids=walk_ids, replay=replay)
        st['seeds'].run(self.seeds_buf)
        return self.emb_all[:b], self.emb_all[b:]

    def _general_layer_train(self, layer):
        m, other = self.model, self.other
        last = layer == self.n_hops - 1
        for t in _TYPES:
            for (s, h, side, r0, n) in self.segments[t]:
                if h > self.n_hops - 1 - layer:
                    continue
                _, c0, cn = self.where[s + 2]
                kernels.gather_mean(self.h[layer - 1][other[t]][c0:c0 + cn], None, n, self.fanouts[h],
                                    out=self.x_mean[layer][t][r0:r0 + n])
        for t in _TYPES:
            agg = m._aggs[layer][t]
            rows = self.h[layer][t].shape[0]
            kernels.sage_project(self.h[layer - 1][t][:rows], self.x_mean[layer][t][:rows], agg.w_self, agg.w_neigh,
                                 agg.bias, agg.activation, out=self.z[t] if last else self.h[layer][t])

    def _g_self_dst(self, layer, t, g_prev):
        """Where the own-row input gradient of (layer, type) goes.  With one head
        segment (the roots, rows [0, B)) that is nobody's neighbor slot, it is the
        roots' part of the previous layer's gradient itself; otherwise a buffer
        that is added to the neighbor contribution afterwards."""
        heads = [(s, h, r0, n) for (s, h, side, r0, n) in self.segments[t] if h <= self.n_hops - 1 - layer]
        if len(heads) == 1 and heads[0][1] == 0 and heads[0][2] == 0:
            return g_prev[t][:heads[0][3]]
        return self.g_xs[layer][t]

    # ---- backward -----------------------------------------------------------------
    def can_update_in_backward(self, grads_flat, head_grads):
        """True when ``backward_head`` can apply the optimizer step inside its reduction: the one-kernel
        backward pass runs on the model's own weights, and the gradients it writes (the aggregators' and
        ``head_grads``) are views of ``grads_flat`` that cover all of it."""
        m = self.model
        relu0 = all(m._aggs[0][t].activation == 'relu' for t in _TYPES)
        two = self.fused_backward and self._tail_kernel and (self.h1_mask is not None or not relu0)
        if not (two or self.fused_backward3):
            return False
        if self._padded_model is not None:
            return False
        gs = [g for layer in range(self.n_hops) for t in _TYPES for g in self._grads_of(m._aggs[layer][t]) if g is not None] + list(head_grads)
        lo, hi = grads_flat.data_ptr(), grads_flat.data_ptr() + grads_flat.numel() * 4
        return (all(lo <= g.data_ptr() and g.data_ptr() + g.numel() * 4 <= hi for g in gs)
                and sum(g.numel() for g in gs) == grads_flat.numel())

    def backward_head(self, head_kernel, head_bias, targets, g_kernel, g_bias, pred, loss, adam=None):
        """The link head (concat -> Dense(1, sigmoid), mean squared error against ``targets``) and the
        whole backward pass from the embeddings ``forward_train`` left in the engine: the head is
        evaluated inside the kernel that starts the backward pass, its parameter gradients and the loss
        are summed beside the weight gradients.  Only when ``tail_kernel``.  ``adam`` (see
        ``kernels.Sage2BackwardAll.run``; only when ``can_update_in_backward``): the optimizer step in the
        same reduction."""
        if self.fused_backward3 and not self.tail_kernel:
            return self._backward_three(None, None, head=(head_kernel, head_bias, targets, g_kernel, g_bias, pred, loss), adam=adam)
        if not self.tail_kernel:
            raise kernels._lib.GsageError("backward_head needs the two-hop one-kernel route or the three-hop one of 32 units per layer")
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._backward_walks(None, None, main, head=(head_kernel, head_bias, targets, g_kernel, g_bias, pred, loss),
                                    adam=adam)

    def _backward_three(self, g_user, g_item, head=None, adam=None):
        """Three hops, 32 units per layer, on the layered forward's tables: a kernel over the seeds, one over the
        hop-1 nodes (each with its children's rows), one reduction; every weight gradient and, with ``head``, the
        link head and the loss."""
        m = self.model
        key = (kernels.storage_epoch(), head is not None, None if head is None else tuple(x.data_ptr() for x in head[:2] + head[3:]),
               self.subtree_train)
        if self._bwd3 is None or self._bwd3[0] != key:
            b, n0, n1 = self.batch, self.fanouts[0], self.fanouts[1]
            rows0, rows1 = b * (1 + n0 + n0 * n1), b * (1 + n0)
            saved = {'z': self.z, 'h0': self.h[0], 'h1': self.h[1], 'row_ids': self.row_ids, 'feat': self.tables,
                     'x_mean0': {t: self.x_mean[0][t][:rows0] for t in _TYPES},
                     'x_mean1': {t: self.x_mean[1][t][:rows1] for t in _TYPES},
                     'x_mean2': {t: self.x_mean[2][t][:b] for t in _TYPES}}
            emb = self.emb
            if self.subtree_train:
                # the rows below the seeds where the subtree kernel left them: the hop-1 nodes' outputs and the ids in
                # arrays over both node types (those below the user seeds -- items -- first), the rest in the tables
                st = self._sub_train
                h1n, h2n = b * n0, b * n0 * n1
                part = lambda x, t, n: x[:n] if t == 'item' else x[n:]          # hop-1 nodes of type t
                part2 = lambda x, t, n: x[:n] if t == 'user' else x[n:]         # hop-2 nodes of type t (below item-typed hop-1 nodes first)
                for name in ('h0', 'h1', 'x_mean0', 'x_mean1'):
                    saved[name] = {t: saved[name][t][:b] for t in _TYPES}
                saved['row_ids'] = {'user': self.seeds_buf[:b], 'item': self.seeds_buf[b:]}
                saved['hop1'] = {'h0': {t: part(st['h0'], t, h1n) for t in _TYPES}, 'h1': {t: part(st['h1'], t, h1n) for t in _TYPES},
                                 'x_mean0': {t: self.x_mean[0][t][b:rows1] for t in _TYPES},
                                 'x_mean1': {t: self.x_mean[1][t][b:rows1] for t in _TYPES},
                                 'ids': {t: part(st['hop1'], t, h1n) for t in _TYPES}}
                saved['hop2'] = {'mask': st['mask'], 'x_mean0': {t: self.x_mean[0][t][rows1:rows0] for t in _TYPES},
                                 'ids': {t: part2(st['hop2'], t, h2n) for t in _TYPES}}
                emb = {'user': self.emb_all[:b], 'item': self.emb_all[b:]}
            weights = {layer: {t: (m._aggs[layer][t].w_self, m._aggs[layer][t].w_neigh) for t in _TYPES} for layer in (1, 2)}
            grads = {(layer, t): self._grads_of(m._aggs[layer][t]) for layer in (0, 1, 2) for t in _TYPES}
            hd = None
            if head is not None:
                head_kernel, head_bias, targets, g_kernel, g_bias, pred, loss = head
                hd = ({'user': emb['user'], 'item': emb['item']}, targets, head_kernel, head_bias, g_kernel, g_bias, pred, loss)
            self._bwd3 = (key, kernels.Sage3BackwardAll(saved, weights, grads, n0, n1, m.normalize == "l2",
                                                        m._aggs[2]['user'].activation, head=hd))
        if head is not None:
            self._bwd3[1].run(targets=head[2], adam=adam)
        else:
            self._bwd3[1].run(g_emb={'user': g_user.contiguous(), 'item': g_item.contiguous()}, adam=adam)
        if self._padded_model is not None:
            self._padded_model.scatter_grads()

    def _backward_walks(self, g_user, g_item, main, head=None, adam=None):
        """Two hops on the one-kernel forward's tables: one launch takes the embedding
        gradients through the l2 normalisation, the layer-1 input gradients and the
        neighbor-mean broadcast for both node types; after it only weight gradients
        are left, and none of them waits for another.  With equal feature widths
        the two node types of a layer share one launch (layer 0 on the caller's
        stream, the small layer 1 beside it); otherwise layer 0 of the user side
        runs on the caller's stream, of the item side on a second, layer 1 on a
        third."""
        m = self.model
        top = m._aggs[1]
        relu0 = all(m._aggs[0][t].activation == 'relu' for t in _TYPES)
        if self.fused_backward and self._tail_kernel and (self.h1_mask is not None or not relu0):
            # everything above in ONE launch, the weight gradients of both layers and the head's with it; one
            # reduction of the blocks' partial sums follows
            key = (kernels.storage_epoch(), head is not None, None if head is None else tuple(x.data_ptr() for x in head[:2] + head[3:]))
            if self._bwd_all is None or self._bwd_all[0] != key:
                rows = {t: self.h[0][t].shape[0] for t in _TYPES}
                saved = {'z': self.z, 'h1': self.h[0], 'x_mean1': {t: self.x_mean[1][t][:self.batch] for t in _TYPES},
                         'x_mean': {t: self.x_mean[0][t][:rows[t]] for t in _TYPES}, 'h1_mask': self.h1_mask}
                if self.self_ids is not None:
                    saved['x_self'], saved['self_ids'] = {t: self.tables[t] for t in _TYPES}, self.self_ids
                else:
                    saved['x_self'] = self.x_self
                grads = {(layer, t): self._grads_of(m._aggs[layer][t]) for layer in (0, 1) for t in _TYPES}
                hd = None
                if head is not None:
                    head_kernel, head_bias, targets, g_kernel, g_bias, pred, loss = head
                    hd = ({'user': self.emb_all[:self.batch], 'item': self.emb_all[self.batch:]}, targets, head_kernel,
                          head_bias, g_kernel, g_bias, pred, loss)
                call = kernels.Sage2BackwardAll(saved, {t: (top[t].w_self, top[t].w_neigh) for t in _TYPES}, grads,
                                                self.fanouts[0], m.normalize == "l2", top['user'].activation, head=hd,
                                                input_scale=1.0 / (1.0 - self.dropout) if self.dropout else 1.0)
                self._bwd_all = (key, call)
            if head is not None:
                self._bwd_all[1].run(targets=head[2], adam=adam)
            else:
                self._bwd_all[1].run(g_emb={'user': g_user.contiguous(), 'item': g_item.contiguous()})
            if self._padded_model is not None:
                self._padded_model.scatter_grads()
            return
        if adam is not None:
            raise kernels._lib.GsageError("the optimizer step inside the backward pass needs its one-kernel form "
                                          "(can_update_in_backward)")
        if self._side2 is None:
            self._side2 = torch.cuda.Stream()
        l0_mask = None if self.h1_mask is None else {'h1': self.h[0], 'h1_mask': self.h1_mask}
        if self.dropout:
            l0_mask['input_scale'] = 1.0 / (1.0 - self.dropout)
        if head is None:
            kernels.sage2_backward_tail(self.z, {'user': g_user.contiguous(), 'item': g_item.contiguous()},
                                        {t: top[t].w_self for t in _TYPES}, {t: top[t].w_neigh for t in _TYPES},
                                        self.fanouts[0], m.normalize == "l2", top['user'].activation, self.g_z, self.g_h[0],
                                        l0_mask=l0_mask)
        else:
            head_kernel, head_bias, targets, g_kernel, g_bias, pred, loss = head
            if self._head_slab is None:
                self._head_slab = kernels.tail_head_slab(self.batch, m.dims[2]['user'], self.indptr.device)
            kernels.sage2_backward_tail_head(self.z, {'user': self.emb_all[:self.batch], 'item': self.emb_all[self.batch:]},
                                             targets, head_kernel, head_bias,
                                             {t: top[t].w_self for t in _TYPES}, {t: top[t].w_neigh for t in _TYPES},
                                             self.fanouts[0], m.normalize == "l2", top['user'].activation,
                                             self.g_z, self.g_h[0], pred, self._head_slab, l0_mask=l0_mask)
        f = {t: self.x_self[t].shape[1] for t in _TYPES}
        if f['user'] == f['item']:
            # both node types of a layer have one shape: each layer's weight gradients are one launch (+ one
            # reduction); layer 0 stays on the caller's stream, the small layer 1 goes beside it
            if self._wgrad_pairs is None or self._wgrad_pairs[0].epoch != kernels.storage_epoch():
                self._wgrad_pairs = [self._wgrad_pair(0), self._wgrad_pair(1)]   # gradient views of the current layout
            self._side2.wait_stream(main)
            self._wgrad_pairs[0].run()
            with torch.cuda.stream(self._side2):
                if head is not None:                        # the head's slab rows, beside the weight gradients
                    kernels.link_head_reduce(self._head_slab, self.batch, m.dims[2]['user'], g_kernel, g_bias, loss)
                self._wgrad_pairs[1].run()
            main.wait_stream(self._side2)
        else:
            self._side.wait_stream(main)
            self._side2.wait_stream(main)
            if head is not None:
                with torch.cuda.stream(self._side2):
                    kernels.link_head_reduce(self._head_slab, self.batch, m.dims[2]['user'], g_kernel, g_bias, loss)
            for t, stream in (('user', main), ('item', self._side)):
                with torch.cuda.stream(stream):
                    self._weight_grads(0, t)
            with torch.cuda.stream(self._side2):
                for t in _TYPES:
                    self._weight_grads(1, t)
            main.wait_stream(self._side)
            main.wait_stream(self._side2)
        if self._padded_model is not None:
            self._padded_model.scatter_grads()

    def _wgrad_inputs(self, layer, t):
        """(own rows, neighbor means, forward output, output gradient) of (layer, type) on the one-kernel
        forward's tables."""
        if layer == 0:
            rows = self.h[0][t].shape[0]
            return self.x_self[t], self.x_mean[0][t][:rows], self.h[0][t], self.g_h[0][t]
        return self.h[0][t][:self.batch], self.x_mean[1][t][:self.batch], self.z[t], self.g_z[t]

    def _grads_of(self, agg):
        for p in agg.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        return agg.w_self.grad, agg.w_neigh.grad, None if agg.bias is None else agg.bias.grad

    def _wgrad_act(self, layer, t):
        """Activation the weight-gradient kernels apply: none where the gradient arrives masked already."""
        return 'linear' if layer == 0 and self.h1_mask is not None else self.model._aggs[layer][t].activation

    def _weight_grads(self, layer, t):
        agg = self.model._aggs[layer][t]
        x_self, x_mean, out, g_out = self._wgrad_inputs(layer, t)
        g_ws, g_wn, g_b = self._grads_of(agg)
        kernels.sage_project_bwd(x_self, x_mean, agg.w_self, agg.w_neigh, self._wgrad_act(layer, t), out, g_out,
                                 need_x_self=False, need_x_mean=False, need_bias=g_b is not None,
                                 g_w_self=g_ws, g_w_neigh=g_wn, g_bias=g_b, workspace=self.ws[layer][t])

    def _wgrad_pair(self, layer):
        problems = []
        for t in _TYPES:
            agg = self.model._aggs[layer][t]
            x_self, x_mean, out, g_out = self._wgrad_inputs(layer, t)
            g_ws, g_wn, g_b = self._grads_of(agg)
            prob = {'x_self': x_self, 'x_mean': x_mean, 'out': out if self._wgrad_act(layer, t) == 'relu' else None,
                    'g_out': g_out, 'g_w_self': g_ws, 'g_w_neigh': g_wn, 'g_bias': g_b, 'workspace': self.ws[layer][t]}
            if layer == 0 and self.self_ids is not None:    # own rows from the feature table, by the saved ids
                prob.update(x_self=self.tables[t][0], self_ids=self.self_ids[t], self_off=self.tables[t][1])
            problems.append(prob)
        return kernels.WgradPairCall(problems, self._wgrad_act(layer, 'user'))

    def backward(self, g_user, g_item):
        """Gradients of every aggregator weight from the embedding gradients; they
        are written into ``param.grad`` (which must exist, e.g. as views of a
        flat bucket) and overwrite what was there.

        Within a layer the two node types do not depend on each other, so their
        kernels -- small, latency-bound launches -- run on two streams (user on
        the caller's, item on a side stream) and meet where one type's gradient
        feeds the other's neighbor slots.  A HIP graph captured around this call
        records the same fork/join."""
        m, other, L = self.model, self.other, self.n_hops
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        if self.tail_kernel:
            return self._backward_walks(g_user, g_item, main)
        if self.fused_backward3:
            return self._backward_three(g_user, g_item)
        streams = {'user': main, 'item': self._side}
        g_emb = {'user': g_user.contiguous(), 'item': g_item.contiguous()}
        g_out = {}
        self._side.wait_stream(main)
        for layer in range(L - 1, -1, -1):
            g_prev = self.g_h[layer - 1] if layer > 0 else None
            if layer == 0 and self.row_ids is not None and L > 1:
                # own rows from the feature tables by id, both node types in one launch (after the two streams met)
                main.wait_stream(self._side)
                if self._wgrad_l0_pair is None or self._wgrad_l0_pair.epoch != kernels.storage_epoch():
                    problems = []
                    for t in _TYPES:
                        agg = m._aggs[0][t]
                        g_ws, g_wn, g_b = self._grads_of(agg)
                        rows = self.h[0][t].shape[0]
                        problems.append({'x_self': self.tables[t][0], 'self_ids': self.row_ids[t], 'self_off': self.tables[t][1],
                                         'x_mean': self.x_mean[0][t][:rows], 'out': self.h[0][t], 'g_out': g_out[t],
                                         'g_w_self': g_ws, 'g_w_neigh': g_wn, 'g_bias': g_b, 'workspace': self.ws[0][t]})
                    self._wgrad_l0_pair = kernels.WgradPairCall(problems, 'relu')
                self._wgrad_l0_pair.run()
                break
            for t in _TYPES:
                with torch.cuda.stream(streams[t]):
                    if layer == L - 1:
                        g_out[t] = (kernels.l2_normalize_bwd(self.z[t], g_emb[t], g_x=self.g_z[t])
                                    if m.normalize == "l2" else g_emb[t])
                    agg = m._aggs[layer][t]
                    rows = self.h[layer][t].shape[0]
                    out = self.z[t] if layer == L - 1 else self.h[layer][t]
                    x_self = self.x_self[t] if layer == 0 else self.h[layer - 1][t][:rows]
                    for p in agg.parameters():
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                    kernels.sage_project_bwd(
                        x_self, self.x_mean[layer][t][:rows], agg.w_self, agg.w_neigh, agg.activation, out, g_out[t],
                        need_x_self=layer > 0, need_x_mean=layer > 0, need_bias=agg.bias is not None,
                        g_w_self=agg.w_self.grad, g_w_neigh=agg.w_neigh.grad,
                        g_bias=None if agg.bias is None else agg.bias.grad,
                        g_x_self=self._g_self_dst(layer, t, g_prev) if layer > 0 else None,
                        g_x_mean=self.g_xm[layer][t] if layer > 0 else None, workspace=self.ws[layer][t])
            if layer == 0:
                break
            # each type's neighbor-slot gradients come from the OTHER type's g_xm: the streams meet here
            main.wait_stream(self._side)
            self._side.wait_stream(main)
            # gradient of h[layer-1]: head contribution (prefix rows) and neighbor contribution
            # (every slot beyond the roots is the neighbor slot of the slot two below it)
            for t in _TYPES:
                with torch.cuda.stream(streams[t]):
                    for (s, h, side, r0, n) in self.segments[t]:
                        if h > L - layer:                      # slot not alive at layer-1's output
                            continue
                        dst = g_prev[t][r0:r0 + n]
                        wrote = False
                        if h >= 1:                             # neighbor slot of slot s-2 (other type)
                            _, p0, pn = self.where[s - 2]
                            kernels.gather_mean_bwd(self.g_xm[layer][other[t]][p0:p0 + pn], None, self.fanouts[h - 1], dst)
                            wrote = True
                        if h <= L - 1 - layer:                 # head slot at this layer: its own-row gradient
                            if wrote:
                                dst.add_(self.g_xs[layer][t][r0:r0 + n])
                            elif self._g_self_dst(layer, t, g_prev) is self.g_xs[layer][t]:
                                dst.copy_(self.g_xs[layer][t][r0:r0 + n])
                            # else: the projection backward wrote the roots' rows in place
            g_out = {t: g_prev[t] for t in _TYPES}
        main.wait_stream(self._side)
        if self._padded_model is not None:
            self._padded_model.scatter_grads()
