"""Static launch plan for the forward hot path (sample -> aggregate -> embed).

``GraphSAGE.__call__`` mirrors the reference layer by layer and allocates as it
goes; this engine is the same arithmetic laid out once for a fixed batch size:
all buffers are allocated up front, slots of one node type share a buffer so a
layer is ONE projection launch per type, and a step is a fixed list of kernel
launches on the current stream with no host synchronisation (it can be captured
in a HIP graph).  It is what the benchmark times and what inference uses.

Three routes, picked at construction: two hops with supported widths run as ONE
kernel (``gs_sage2_forward``, one wavefront per walk); other depths with
supported widths as the sampler kernel plus one fused kernel per (layer, node
type); everything else (any width, the max aggregator) as sampler + gathers +
projections, with the item type's gather-means on a second stream beside the
user type's projection.

Buffer layout.  Slot ``s = 2*h + side`` (hop ``h`` of the walks rooted at head
``side``) has ``B * P_h`` nodes of type ``TYPES[(side + h) % 2]``.  Per node
type ``T`` the head rows of all alive slots of that type are stored back to
back in slot order: ``x_self[T]`` (own features), ``x_mean[T]`` (mean of the
sampled neighbors' features, which are of the other type) and the layer outputs
``h[l][T]``.  Slots die from the deepest hop upwards as layers are applied, so
the rows alive at layer ``l`` are always a prefix of the buffer.
"""
import torch

from . import kernels

_TYPES = ('user', 'item')


class SageForwardEngine:
    def __init__(self, graph, model, batch_size, use_fused=True, walk_kernel=True, keep_ids=False, draw_in_layer=False,
                 wide_min_rows=128 * 256, subtree_kernel=True):
        """draw_in_layer: the layer-0 kernel draws the deepest hop's ids itself instead of reading what
        the sampler kernel wrote (``gs_sage_fused_sampled_fwd``); keep_ids then also leaves those ids
        in ``hops[-1]``.  Off by default: a draw costs one 64-byte sector of the neighbor array wherever
        it is made, and on a graph whose neighbor array exceeds the caches the sampler kernel, which
        does nothing else, moves those sectors faster (measured on the 1M x 100k graph, fan-outs
        25/10/5: 765 us per step against 872 us)."""
        self.graph = graph
        self.keep_ids = bool(keep_ids)
        self.model = model.build()
        self.fanouts = [int(n) for n in model.n_samples]
        self.n_hops = len(self.fanouts)
        self.batch = int(batch_size)
        dev = graph.device_tensors()
        self.tables = {'user': (dev['user'], 0), 'item': (dev['item'], graph.correction_constant)}
        self.indptr, self.indices = dev['indptr'], dev['indices']
        device = self.indptr.device
        b = self.batch
        sizes = [1] + kernels.hop_sizes(self.fanouts)              # nodes per head at hop h
        self.sizes = sizes

        # segments: for each type, [(slot, hop, side, first_row, n_rows)] over hops 0..L-1
        self.segments = {t: [] for t in _TYPES}
        rows = {t: 0 for t in _TYPES}
        self.where = {}
        for h in range(self.n_hops):
            for side in range(2):
                t = _TYPES[(side + h) % 2]
                n = b * sizes[h]
                self.segments[t].append((2 * h + side, h, side, rows[t], n))
                self.where[2 * h + side] = (t, rows[t], n)
                rows[t] += n
        self.rows = rows
        f = graph.node_feature_sizes
        other = {'user': 'item', 'item': 'user'}
        self.other = other
        new = lambda r, c: torch.empty((r, c), dtype=torch.float32, device=device)
        self.x_self = {t: new(rows[t], f[t]) for t in _TYPES}
        self.x_mean = [{t: new(rows[t], model.dims[0][other[t]]) for t in _TYPES}]
        self.h = []
        for layer in range(self.n_hops):
            alive = {t: sum(n for (_, hh, _, _, n) in self.segments[t] if hh <= self.n_hops - 1 - layer) for t in _TYPES}
            self.h.append({t: new(alive[t], model.dims[layer + 1][t]) for t in _TYPES})
            if layer + 1 < self.n_hops:
                nxt = {t: sum(n for (_, hh, _, _, n) in self.segments[t] if hh <= self.n_hops - 2 - layer) for t in _TYPES}
                self.x_mean.append({t: new(nxt[t], model.dims[layer + 1][other[t]]) for t in _TYPES})
        self.emb = {t: new(b, model.dims[-1][t]) for t in _TYPES}
        self.hops = [torch.empty((2 * b, s), dtype=torch.int32, device=device) for s in sizes[1:]]
        self.err_flag = torch.zeros(2, dtype=torch.int32, device=device)
        self._hop_ptrs = None
        self._side = None            # second stream (overlaps one type's gathers with the other's projection)
        self.probes = None           # list to receive the LaunchTimers of the dominant launches
        self.seeds_buf = torch.zeros(2 * b, dtype=torch.int32, device=device)   # placeholder roots
        # the fused kernels implement the mean aggregator; 'max' runs gather + reduce + projection
        self.agg_max = getattr(model, 'agg_option', 'mean') == 'max'
        use_fused = bool(use_fused) and not self.agg_max
        self._reduce = kernels.gather_max if self.agg_max else kernels.gather_mean
        self.use_fused = use_fused
        self.fused_layers = [self._fused_ok(layer) and use_fused for layer in range(self.n_hops)]
        self._fused_calls = {}
        # Wide layers (half output width a multiple of 128) go through the kernel that overlaps the
        # gathers with the projection.  It pays where the gathers are large, i.e. in layer 0; the later
        # layers have few rows, where a launch of one 128-row tile per compute unit is mostly latency,
        # and run on the general kernels.
        self.wide_first_layer_only = (use_fused and self.n_hops > 1 and
                                      all(model.dims[1][t] // 2 >= 128 for t in _TYPES))
        # (forward only: a later layer with at least ``wide_min_rows`` rows, a 128-row tile per compute
        # unit -- layer 1 of a three-hop model -- takes the wide kernel as well)
        self.wide_layers = [False] * self.n_hops
        if self.wide_first_layer_only and all(self.fused_layers):
            self.wide_layers = [layer == 0 or min(x.shape[0] for x in self.h[layer].values()) >= int(wide_min_rows)
                                for layer in range(self.n_hops)]
        # two hops: the whole forward pass as one kernel, one wavefront per walk
        self.walk_kernel = bool(use_fused and walk_kernel and self.n_hops == 2 and kernels.sage2_supported(
            f['user'], f['item'], model.dims[1]['user'], model.dims[2]['user'], *self.fanouts))
        # Feature widths the one-kernel forward is not built for (MovieLens: 28 / 18 columns) are
        # rounded up to the next multiple of 32 with zero columns in engine-owned copies of the
        # tables, and the layer-0 weights get zero rows to match: the same function of the real
        # features.  ``static_weights``: the padded weight copies are refreshed on every call unless
        # the caller promises the weights do not change between calls.
        self._padded = None
        self.static_weights = False
        if use_fused and walk_kernel and self.n_hops == 2 and not self.walk_kernel:
            fp = {t: -(-f[t] // 32) * 32 for t in _TYPES}
            if kernels.sage2_supported(fp['user'], fp['item'], model.dims[1]['user'], model.dims[2]['user'],
                                       *self.fanouts):
                tabs, wts = {}, {}
                for t in _TYPES:
                    src = self.tables[t][0]
                    tabs[t] = torch.zeros((src.shape[0], fp[t]), dtype=torch.float32, device=device)
                    tabs[t][:, :f[t]].copy_(src)
                    hh = model.dims[1][t] // 2
                    wts[(t, 'self')] = torch.zeros((fp[t], hh), dtype=torch.float32, device=device)
                    wts[(t, 'neigh')] = torch.zeros((fp[other[t]], hh), dtype=torch.float32, device=device)
                self._padded = {'tables': tabs, 'weights': wts, 'fresh': False}
                self.walk_kernel = True
        self._walk_call = None
        # Three hops at the widths of the one-kernel forward: two launches.  The same kernel below the seeds -- a
        # wavefront takes eight hop-1 nodes at a time, draws them and their subtrees' hops 2 and 3 and computes
        # layers 0 and 1 per subtree -- then one kernel for layers 0, 1, 2 at the seeds.  (cfg-2 graph,
        # [25,10,5]: 0.62 ms per step against 0.76 ms for the sampler + one kernel per layer and type.)
        self.subtree_kernel = bool(
            subtree_kernel and use_fused and walk_kernel and self.n_hops == 3 and all(self.fused_layers) and not self.wide_first_layer_only
            and all(model.dims[l]['user'] == model.dims[l]['item'] for l in (1, 2, 3))
            and model._aggs[2]['user'].activation == model._aggs[2]['item'].activation
            and kernels.sage2_subtrees_supported(f['user'], f['item'], model.dims[1]['user'], model.dims[2]['user'], *self.fanouts[1:])
            and kernels.sage3_seeds_supported(f['user'], f['item'], model.dims[1]['user'], model.dims[2]['user'],
                                              model.dims[3]['user'], self.fanouts[0]))
        self._subtree = None
        self.emb_all = torch.empty((2 * b, model.dims[-1]['user']), dtype=torch.float32, device=device)
        # On request (draw_in_layer) the deepest hop -- most of a walk's samples -- is drawn by the
        # layer-0 kernel where its feature rows are gathered, and the sampler kernel stops one hop short.
        # (The wide-layer kernel reads ids, so wide models always sample every hop up front.)
        self.draw_deepest = bool(draw_in_layer and use_fused and all(self.fused_layers) and not self.walk_kernel
                                 and not self.subtree_kernel
                                 and not self.wide_first_layer_only
                                 and all(model.dims[1][t] // 2 in (16, 32) for t in _TYPES))
        self._step_keys = (0, None, None)

    def _arm_probe(self):
        """The next dominant-kernel launch carries a pair of HIP events in its own
        dispatch (exact start and end, nothing extra enqueued); the timer joins
        ``self.probes`` and is read with ``elapsed_us()`` after the run."""
        t = kernels.LaunchTimer()
        t.arm()
        self.probes.append(t)

    # -- pieces ---------------------------------------------------------------
    def _slot_ids(self, seeds, h, side):
        b = self.batch
        src = seeds if h == 0 else self.hops[h - 1]
        return src[side * b:(side + 1) * b].reshape(-1)

    def sample(self, seeds, walk_ids=None, rng_seed=0, replay=None):
        import ctypes
        from ._lib import check, load, ptr, stream_handle
        lib = load()
        if self._hop_ptrs is None:
            self._fan_arr = (ctypes.c_int32 * self.n_hops)(*self.fanouts)
            self._hop_ptrs = (ctypes.c_void_p * self.n_hops)(*[t.data_ptr() for t in self.hops])
        if seeds.numel() != 2 * self.batch or seeds.dtype != torch.int32 or not seeds.is_cuda:
            raise ValueError("seeds must be an int32 GPU tensor of 2*batch roots (sources, then destinations)")
        if walk_ids is not None and (walk_ids.numel() != seeds.numel() or walk_ids.dtype != torch.int64):
            raise ValueError("walk_ids must be int64, one per seed")
        if replay is not None and replay.numel() != seeds.numel() * sum(self.sizes[1:]):
            raise ValueError("replay tape has the wrong length")
        self._step_keys = (rng_seed, walk_ids, replay)
        n_hops = self.n_hops
        if self.draw_deepest:
            n_hops -= 1                                      # the layer-0 kernel draws the last hop
            if n_hops == 0:
                return
            if replay is not None:                           # the sampler strides the tape by ITS hops' words
                t_all, t_part = sum(self.sizes[1:]), sum(self.sizes[1:n_hops + 1])
                replay = replay.reshape(seeds.numel(), t_all)[:, :t_part].contiguous().reshape(-1)
        check(lib.gs_sample(ptr(self.indptr), ptr(self.indices), self.indptr.numel() - 1, ptr(seeds),
                            seeds.numel(), self._fan_arr, n_hops, int(rng_seed) & 0xFFFFFFFFFFFFFFFF, 0,
                            ptr(walk_ids), ptr(replay), self._hop_ptrs, ptr(self.err_flag), stream_handle()),
              'gs_sample')

    def aggregate(self, seeds):
        """Layers 0..L-1 on the ids left by ``sample``; returns (user_emb, item_emb)."""
        m = self.model
        other = self.other
        # Layer 0, per type: the neighbor means of every alive slot (children rows summed as they
        # are fetched from the other type's table), then the projection, which reads the own rows
        # from the table by id -- no gathered copy of either is written.  The means are bound by
        # HBM and the wide projections by the matrix pipe, so the item type's means run on a
        # second stream beside the user type's projection.
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        for t in _TYPES:                                   # 'user' first
            with torch.cuda.stream(main if t == 'user' else self._side):
                feat_o, off_o = self.tables[other[t]]
                for (s, h, side, r0, n) in self.segments[t]:
                    dst = self.x_mean[0][t][r0:r0 + n]
                    probe = self.probes is not None and t == 'user' and h + 1 == self.n_hops   # runs alone
                    if probe:
                        self._arm_probe()
                    self._reduce(feat_o, self._slot_ids(seeds, h + 1, side), n, self.fanouts[h], off_o, out=dst)
                if t == 'user':
                    self._side.wait_stream(main)           # the side stream starts once these means are done
                agg = m._aggs[0][t]
                feat, off = self.tables[t]
                for (s, h, side, r0, n) in self.segments[t]:
                    kernels.sage_project_gather(feat, self._slot_ids(seeds, h, side), off, self.x_mean[0][t][r0:r0 + n],
                                                agg.w_self, agg.w_neigh, agg.bias, agg.activation,
                                                out=self.h[0][t][r0:r0 + n])
        main.wait_stream(self._side)
        return self._aggregate_tail()

    def _general_layer(self, layer):
        """Layer ``layer`` >= 1 with the general kernels (neighbor means to memory, then the tiled projection)."""
        m = self.model
        other = self.other
        for t in _TYPES:
            for (s, h, side, r0, n) in self.segments[t]:
                if h > self.n_hops - 1 - layer:
                    continue
                _, c0, cn = self.where[s + 2]
                self._reduce(self.h[layer - 1][other[t]][c0:c0 + cn], None, n, self.fanouts[h],
                             out=self.x_mean[layer][t][r0:r0 + n])
        for t in _TYPES:
            agg = m._aggs[layer][t]
            rows = self.h[layer][t].shape[0]
            kernels.sage_project(self.h[layer - 1][t][:rows], self.x_mean[layer][t][:rows], agg.w_self, agg.w_neigh,
                                 agg.bias, agg.activation, out=self.h[layer][t])

    def _normalize_general(self):
        for t in _TYPES:
            if self.model.normalize == "l2":
                kernels.l2_normalize(self.h[-1][t], out=self.emb[t])
            else:
                self.emb[t].copy_(self.h[-1][t])
        return self.emb['user'], self.emb['item']

    def _aggregate_tail(self):
        """Layers 1..L-1 and the normalisation with the general kernels, on layer 0's output."""
        for layer in range(1, self.n_hops):
            self._general_layer(layer)
        return self._normalize_general()

    # -- fused layers ----------------------------------------------------------------
    def _fused_ok(self, layer):
        m = self.model
        for t in _TYPES:
            fs, fn, h = m.dims[layer][t], m.dims[layer][self.other[t]], m.dims[layer + 1][t]
            if not kernels.fused_supported(fs, fn, h, max(self.fanouts)):
                return False
        return True

    def _fused_call(self, layer, t):
        """Prepared one-kernel layer for head type ``t`` (built on first use)."""
        key = (layer, t)
        if key in self._fused_calls:
            return self._fused_calls[key]
        m, other = self.model, self.other
        agg = m._aggs[layer][t]
        last = layer == self.n_hops - 1
        segs = self._layer_segments(layer, t)
        if layer == 0:
            (f_self, o_self), (f_neigh, o_neigh) = self.tables[t], self.tables[other[t]]
        else:
            f_self, o_self, f_neigh, o_neigh = self.h[layer - 1][t], 0, self.h[layer - 1][other[t]], 0
        normalize = last and m.normalize == "l2"
        out = self.emb[t] if last else self.h[layer][t]
        call = kernels.FusedLayerCall(f_self, o_self, f_neigh, o_neigh, segs, agg.w_self, agg.w_neigh, agg.bias,
                                      agg.activation, normalize, out,
                                      sampling=self._sampling_desc() if layer == 0 else None)
        self._fused_calls[key] = call
        return call

    def _sampling_desc(self):
        if not self.draw_deepest:
            return None
        return {'indptr': self.indptr, 'indices': self.indices, 'err_flag': self.err_flag,
                'tape_len': sum(self.sizes[1:])}

    def _layer_segments(self, layer, t):
        """Segment table of the fused kernel of (layer, head type ``t``): layer 0 addresses feature
        rows by node id, later layers the previous layer's output rows.  The deepest heads' children
        are drawn in the kernel when ``draw_deepest``."""
        segs = []
        for (s, h, side, r0, n) in self.segments[t]:
            if h > self.n_hops - 1 - layer:
                continue
            if layer == 0:
                sg = {'head_ids': self._slot_ids(self.seeds_buf, h, side), 'n_heads': n, 'n_child': self.fanouts[h]}
                if self.draw_deepest and h == self.n_hops - 1:
                    sg['draw'] = {'hop': h, 'heads_per_walk': self.sizes[h], 'tape_off': sum(self.sizes[1:h + 1]),
                                  'walk0': side * self.batch,
                                  'out_child_ids': self._slot_ids(self.seeds_buf, h + 1, side) if self.keep_ids else None}
                else:
                    sg['child_ids'] = self._slot_ids(self.seeds_buf, h + 1, side)
                segs.append(sg)
            else:
                _, c0, _cn = self.where[s + 2]
                segs.append({'n_heads': n, 'n_child': self.fanouts[h], 'self_row0': r0, 'child_row0': c0})
        return segs

    def _run_fused_layer(self, layer, seeds):
        if self.wide_layers[layer] and self.probes is None:
            # Wide layers: the two node types' launches on two streams.  Each fills the GPU by itself (one
            # block per compute unit), so they do not share it; but the second starts on the compute units the
            # first one's last tiles leave idle (cfg 5: 4.51 -> 4.37 ms per step).  Half-size grids side by side
            # were measured slower (5.2 ms).  A launch that is being timed runs alone.
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(main)
            for t in _TYPES:
                call = self._fused_call(layer, t)
                if layer == 0:
                    for i, (_, h, side, _, _) in enumerate(self.segments[t]):
                        if h == 0:
                            call.set_head_ids(i, self._slot_ids(seeds, 0, side))
                with torch.cuda.stream(main if t == 'user' else self._side):
                    call.run()
            main.wait_stream(self._side)
            return
        for t in _TYPES:
            call = self._fused_call(layer, t)
            if layer == 0:
                # the roots are the only ids that change address from step to step
                for i, (_, h, side, _, _) in enumerate(self.segments[t]):
                    if h == 0:
                        call.set_head_ids(i, self._slot_ids(seeds, 0, side))
            if self.probes is not None and layer == 0:
                self._arm_probe()
            if layer == 0 and self.draw_deepest:
                rng_seed, walk_ids, replay = self._step_keys
                call.run(rng_seed=rng_seed, walk_ids=walk_ids, replay=replay)
            else:
                call.run()

    def aggregate_fused(self, seeds):
        """Same result as ``aggregate`` with one kernel per (layer, node type)."""
        if not all(self.fused_layers):
            return self.aggregate(seeds)
        if self.wide_first_layer_only:
            for layer in range(self.n_hops):
                if self.wide_layers[layer]:
                    self._run_fused_layer(layer, seeds)
                else:
                    self._general_layer(layer)
            if self.wide_layers[-1]:
                return self.emb['user'], self.emb['item']
            return self._normalize_general()
        for layer in range(self.n_hops):
            self._run_fused_layer(layer, seeds)
        return self.emb['user'], self.emb['item']

    def _private_padded(self):
        """A second set of padded layer-0 weights over the same padded tables (a pipeline slot refreshes and reads
        its own copy on its own stream); None when the engine does not pad."""
        if self._padded is None:
            return None
        return {'tables': self._padded['tables'], 'weights': {k: torch.zeros_like(w) for k, w in self._padded['weights'].items()},
                'fresh': False}

    def _refresh_padded_weights(self, pad=None):
        """Copies the real layer-0 weights into the zero-padded ones, on the CURRENT stream: the launch that reads
        them must be ordered behind this call on that stream."""
        m = self.model
        pad = self._padded if pad is None else pad
        if pad is not None and not (self.static_weights and pad['fresh']):
            with torch.no_grad():
                for t in _TYPES:
                    agg = m._aggs[0][t]
                    pad['weights'][(t, 'self')][:agg.w_self.shape[0]].copy_(agg.w_self)
                    pad['weights'][(t, 'neigh')][:agg.w_neigh.shape[0]].copy_(agg.w_neigh)
            pad['fresh'] = True

    def _make_walk_call(self, emb, pad=None):
        """A prepared one-kernel forward that writes its embeddings to ``emb`` (2*batch rows); ``pad``: the set of
        padded weights it reads (default: the engine's own)."""
        m = self.model
        pad = self._padded if pad is None else pad
        weights = {(l, t): (m._aggs[l][t].w_self, m._aggs[l][t].w_neigh, m._aggs[l][t].bias)
                   for l in (0, 1) for t in _TYPES}
        tables = {t: self.tables[t][0] for t in _TYPES}
        if pad is not None:
            tables = pad['tables']
            for t in _TYPES:
                weights[(0, t)] = (pad['weights'][(t, 'self')], pad['weights'][(t, 'neigh')], m._aggs[0][t].bias)
        return kernels.Sage2Call(self.indptr, self.indices, tables['user'], tables['item'],
                                 self.fanouts, weights, m.normalize == "l2", self.batch, emb=emb)

    def forward_walks(self, seeds, walk_ids=None, rng_seed=0, replay=None):
        """sample + both layers + l2norm in one launch (two-hop models)."""
        self._refresh_padded_weights()
        if self._walk_call is None:
            self._walk_call = self._make_walk_call(self.emb_all)
        if self.probes is not None:
            self._arm_probe()
        self._walk_call.run(seeds, self.err_flag, rng_seed=rng_seed, walk_ids=walk_ids, replay=replay)
        return self.emb_all[:self.batch], self.emb_all[self.batch:]

    def pipeline(self, depth=2):
        """Several independent batches in flight (inference over a stream of batches): see ForwardPipeline."""
        return ForwardPipeline(self, depth)

    def _make_subtree_state(self, hop1, emb, deeper_hops=None):
        """The prepared calls of the two-launch three-hop route with their private buffers: the hop-1 ids go to
        ``hop1``, the embeddings to ``emb`` (2*batch rows)."""
        m, b = self.model, self.batch
        n0, n1, n2 = self.fanouts
        dev = self.indptr.device
        h1w, h2w = m.dims[1]['user'], m.dims[2]['user']
        st = {'h0': torch.empty((2 * b * n0, h1w), dtype=torch.float32, device=dev),       # layer 0 at the hop-1 nodes
              'h1': torch.empty((2 * b * n0, h2w), dtype=torch.float32, device=dev)}       # layer 1 at the hop-1 nodes
        weights = {(l, t): (m._aggs[l][t].w_self, m._aggs[l][t].w_neigh, m._aggs[l][t].bias)
                   for l in (0, 1, 2) for t in _TYPES}
        st['call'] = kernels.Sage2Call(
            self.indptr, self.indices, self.tables['user'][0], self.tables['item'][0], [n1, n2],
            {k: w for k, w in weights.items() if k[0] < 2}, False, b,
            emb=st['h1'], hop_buffers=deeper_hops,
            subtrees={'roots_per_walk': n0, 'tape_len': sum(self.sizes[1:]), 'tape_off_hop2': self.sizes[1],
                      'tape_off_hop3': self.sizes[1] + self.sizes[2], 'out_h0': st['h0'],
                      'draw_hop1': hop1.reshape(-1)})
        st['seeds'] = kernels.Sage3SeedsCall(
            self.tables['user'][0], self.tables['item'][0], hop1.reshape(-1), n0, st['h0'], st['h1'], weights,
            m._aggs[2]['user'].activation, m.normalize == "l2", b, emb)
        return st

    def forward_subtrees(self, seeds, walk_ids=None, rng_seed=0, replay=None):
        """Three hops in two launches: everything below the seeds in one kernel (all three hops drawn, layers 0 and
        1 computed per hop-1 node's subtree), then layers 0, 1, 2 at the seeds in a second one."""
        m, b, other = self.model, self.batch, self.other
        n0, n1, n2 = self.fanouts
        if seeds.numel() != 2 * b or seeds.dtype != torch.int32 or not seeds.is_cuda:
            raise ValueError("seeds must be an int32 GPU tensor of 2*batch roots (sources, then destinations)")
        if walk_ids is not None and (walk_ids.numel() != seeds.numel() or walk_ids.dtype != torch.int64):
            raise ValueError("walk_ids must be int64, one per seed")
        tape_len = sum(self.sizes[1:])
        if replay is not None and replay.numel() != seeds.numel() * tape_len:
            raise ValueError("replay tape has the wrong length")
        if self._subtree is None:
            self._subtree = self._make_subtree_state(self.hops[0], self.emb_all,
                                                     (self.hops[1], self.hops[2]) if self.keep_ids else None)
        st = self._subtree
        if self.probes is not None:
            self._arm_probe()
        st['call'].run(seeds, self.err_flag, rng_seed=rng_seed, walk_ids=walk_ids, replay=replay)
        st['seeds'].run(seeds)
        return self.emb_all[:b], self.emb_all[b:]

    def forward(self, seeds, walk_ids=None, rng_seed=0, replay=None):
        if self.walk_kernel:
            return self.forward_walks(seeds, walk_ids, rng_seed, replay)
        if self.subtree_kernel:
            return self.forward_subtrees(seeds, walk_ids, rng_seed, replay)
        self.sample(seeds, walk_ids, rng_seed, replay)
        if self.use_fused and all(self.fused_layers):
            return self.aggregate_fused(seeds)
        return self.aggregate(seeds)

    # -- accounting for the benchmark --------------------------------------------
    def dominant_name(self):
        if self.walk_kernel:
            return "sage2_kernel (sample + both layers + l2norm, one wavefront per walk)"
        if self.subtree_kernel:
            return ("sage2_kernel below the seeds (all three hops drawn, layers 0 and 1 of every hop-1 node's "
                    "subtree, eight subtrees per wavefront at a time)")
        if self.use_fused and all(self.fused_layers):
            if self.wide_first_layer_only:
                return ("sage_wide_kernel (layer 0: neighbor gather + mean on producer waves beside the MFMA "
                        "projection, one launch per node type)")
            return "sage_fused_kernel (layer 0: gather + neighbor mean + projection, one launch per node type)"
        return "gather_mean_kernel (deepest hop: rows summed as fetched)"

    def dominant_flops(self):
        """Multiply-add flops of ONE launch of the dominant kernel where it is a
        matrix product worth pricing (the wide layer-0 kernel: average of the two
        node types' launches); None otherwise."""
        if not (self.use_fused and all(self.fused_layers) and self.wide_first_layer_only) or self.walk_kernel:
            return None
        f = self.graph.node_feature_sizes
        per_type = [sum(n for (_, _, _, _, n) in self.segments[t]) * (f[t] + f[self.other[t]]) * self.model.dims[1][t]
                    for t in _TYPES]                     # rows * (Fs + Fn) * H/2 * 2
        return sum(per_type) / len(per_type)

    def dominant_bytes(self):
        """Algorithmic bytes of ONE launch of the deepest-hop gather-mean: child
        ids read, one feature row read per child, one mean row written per
        parent (average of the user-table and item-table launches)."""
        b, h = self.batch, self.n_hops - 1
        f = self.graph.node_feature_sizes
        if self.walk_kernel:
            # the one-kernel forward: CSR reads as the sampler's accounting (without the id
            # round trip), every sampled node's feature row once, one embedding row per root
            sampler, agg = self._layered_bytes()
            samples = 2 * sum(self.sizes[1:])
            rows = sum(b * self.sizes[hh] * f[_TYPES[(side + hh) % 2]] * 4
                       for hh in range(self.n_hops + 1) for side in range(2))
            return (sampler - b * 4 * samples) + rows + 2 * b * self.model.dims[-1]['user'] * 4
        if self.subtree_kernel:
            # per hop-1 node: its id, the CSR reads of its subtree's draws (row bounds of 1 + n1 expanded nodes, one
            # neighbor word per draw), every subtree node's feature row once, its two output rows
            # (its own draw: one neighbor word read, the id written; per seed walk the seed id and its row bounds)
            n0, n1, n2 = self.fanouts
            per_root = {}
            for side, t in enumerate(_TYPES):               # t = type of the subtree's root
                o = self.other[t]
                rows = f[t] * 4 * (1 + n1 * n2) + f[o] * 4 * n1
                per_root[t] = 8 + 16 * (1 + n1) + 4 * (n1 + n1 * n2) + rows + 4 * (self.model.dims[1][t] + self.model.dims[2][t])
            return b * n0 * sum(per_root.values()) + 2 * b * (4 + 16)
        if self.use_fused and all(self.fused_layers):
            # fused layer 0, one launch per head type: head + child ids read, one feature row per
            # head and per child read, one output row per head written
            per_type = []
            for t in _TYPES:
                tot = 0
                for (_, hh, _, _, n) in self.segments[t]:
                    kids = n * self.fanouts[hh]
                    tot += (n + kids) * 4 + n * f[t] * 4 + kids * f[self.other[t]] * 4
                    tot += n * self.model.dims[1][t] * 4
                per_type.append(tot)
            return sum(per_type) / len(per_type)
        parents, children = b * self.sizes[h], b * self.sizes[h + 1]
        per_type = [children * 4 + children * f[t] * 4 + parents * f[t] * 4 for t in _TYPES]
        return sum(per_type) / len(per_type)

    def measured_traffic(self, workload):
        """HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes,
        if a measurement of this kernel on this workload (graph, batch, fan-outs)
        is stored under profiles/ AND was taken on the kernel source that is in the
        tree now (the entry carries the git blob hash of the kernel's file); None
        otherwise, so that a stale figure is never printed."""
        import hashlib
        import json
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, 'profiles', 'traffic.json')
        if not os.path.exists(path):
            return None
        with open(path) as fh:
            rec = json.load(fh)
        name = "sage_fused_kernel" if self.use_fused and all(self.fused_layers) else "gather_mean_kernel"
        if name == "sage_fused_kernel" and self.wide_first_layer_only:
            name = "sage_wide_kernel"
        if self.walk_kernel or self.subtree_kernel:
            name = "sage2_kernel"
        tail = "/B%d/%s" % (self.batch, 'x'.join(str(n) for n in self.fanouts))
        # entries are keyed by the kernel's full name, template arguments included
        if name == "sage_fused_kernel":          # the layer-0 instantiation: <Fs, Fn, H/2, ids, ...>; both node types, averaged
            f = self.graph.node_feature_sizes
            hits = [v for k, v in rec.items() if k.endswith(tail) and any(
                k.startswith("%s/sage_fused_kernel<%d, %d, %d, false" % (workload, f[t], f[self.other[t]], self.model.dims[1][t] // 2))
                for t in _TYPES)]
        elif name == "sage2_kernel":             # <Fu, Fi, H1/2, H2/2, buffer loads, SAVING, SUBTREES, DROPOUT>: the forward saves nothing
            flags = "false, true, false>" if self.subtree_kernel else "false, false, false>"
            hits = [v for k, v in rec.items() if k.startswith("%s/sage2_kernel<" % workload) and k.endswith(flags + tail)]
        else:
            hits = [v for k, v in rec.items() if k.startswith("%s/%s<" % (workload, name)) and k.endswith(tail)]
        out = []
        for entry in hits:
            if not isinstance(entry, dict):
                return None                       # an entry without a source hash cannot be trusted
            for src, want in entry.get('source', {}).items():
                try:
                    data = open(os.path.join(root, src), 'rb').read()
                except OSError:
                    return None
                if hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest() != want:
                    return None                   # the kernel has changed since it was measured
            out.append(float(entry['bytes']))
        return sum(out) / len(out) if out else None

    # algorithmic bytes of one step, per the accounting used for the roofline
    def algorithmic_bytes(self):
        """(sampler bytes, gather+aggregate bytes) of one batch by the layered accounting, whatever route the
        engine takes: ids and CSR row bounds read, sampled ids written; every sampled node's feature row read
        once and the layer-0 inputs (own row + neighbor mean per head node) written once.  The one- and
        two-launch routes move less (no id round trip, no layer inputs written): ``dominant_bytes`` is their own
        accounting, and the one the roofline figure uses."""
        return self._layered_bytes()

    def _layered_bytes(self):
        b = self.batch
        f = self.graph.node_feature_sizes
        sizes = self.sizes
        expansions = 2 * sum(sizes[h] for h in range(self.n_hops))
        samples = 2 * sum(sizes[h] for h in range(1, self.n_hops + 1))
        sampler = b * (8 + 16 * expansions + 4 * samples + 4 * samples)
        rows_read = 0
        written = 0
        for h in range(self.n_hops + 1):
            for side in range(2):
                t = _TYPES[(side + h) % 2]
                rows_read += b * sizes[h] * f[t] * 4
                if h < self.n_hops:
                    written += b * sizes[h] * (f[t] + f[self.other[t]]) * 4
        return sampler, rows_read + written


class ForwardPipeline:
    """``depth`` independent batches of the one-kernel forward (two hops) or of the two-launch forward (three
    hops) in flight, each on a stream of its own with its own output rows.

    A launch of the one-kernel forward does not keep the whole chip busy to its end: the walks rooted at items
    finish well before those rooted at users (their deepest rows come from the smaller table), and within a side
    the compute units drain one by one.  A second batch on a second stream starts on the compute units the first
    one frees (cfg-2 graph, 4096 seed edges per batch: 78 us per batch against 94 us one at a time).  Batches of
    a training loop depend on each other through the weights and cannot be run this way; scoring a stream of
    links can.

    ``submit`` returns the (user, item) embedding views of the slot it used and an event recorded behind the
    launch; the caller reads them (after ``event.synchronize()`` or ``torch.cuda.current_stream().wait_event``)
    before ``depth`` further batches have been submitted, because the slot's rows are then written again.  The
    batch's input tensors are read on the slot's stream: they must stay allocated until that event has passed.
    """

    def __init__(self, engine, depth=2):
        if not (engine.walk_kernel or engine.subtree_kernel):
            raise kernels._lib.GsageError("the forward pipeline runs the one-kernel (two hops) and two-launch (three "
                                          "hops) routes")
        if depth < 1:
            raise ValueError("depth must be at least 1")
        self.engine = engine
        self.depth = int(depth)
        dev = engine.indptr.device
        self._emb = [torch.empty_like(engine.emb_all) for _ in range(self.depth)]
        if engine.walk_kernel:
            # odd feature widths: every slot keeps padded layer-0 weights of its own, refreshed on the slot's stream
            # right before its launch, so no copy ever runs beside a launch that reads its destination
            self._pads = [engine._private_padded() for _ in range(self.depth)]
            self._calls = [engine._make_walk_call(e, pad) for e, pad in zip(self._emb, self._pads)]
        else:                                                    # every slot: its own hop-1 ids and hop-1 activations
            self._hop1 = [torch.empty_like(engine.hops[0]) for _ in range(self.depth)]
            self._states = [engine._make_subtree_state(h, e) for h, e in zip(self._hop1, self._emb)]
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.depth)]
        self._events = [torch.cuda.Event() for _ in range(self.depth)]
        self._n = 0

    def submit(self, seeds, walk_ids=None, rng_seed=0, replay=None, inputs_ready=False):
        """``inputs_ready``: the batch's tensors AND the model's weights are complete already (resident since an
        earlier synchronisation), so the slot's stream need not wait for the current stream."""
        eng, b = self.engine, self.engine.batch
        slot = self._n % self.depth
        self._n += 1
        stream = self._streams[slot]
        if not inputs_ready:
            stream.wait_stream(torch.cuda.current_stream())      # the batch's inputs were produced there
        with torch.cuda.stream(stream):
            if eng.walk_kernel:
                # (the slot's previous launch, the only other user of these copies, is ahead on this stream)
                eng._refresh_padded_weights(self._pads[slot])
                self._calls[slot].run(seeds, eng.err_flag, rng_seed=rng_seed, walk_ids=walk_ids, replay=replay)
            else:
                st = self._states[slot]
                st['call'].run(seeds, eng.err_flag, rng_seed=rng_seed, walk_ids=walk_ids, replay=replay)
                st['seeds'].run(seeds)
            self._events[slot].record(stream)
        emb = self._emb[slot]
        return emb[:b], emb[b:], self._events[slot]

    def drain(self):
        """The current stream waits for every batch submitted so far."""
        cur = torch.cuda.current_stream()
        for st in self._streams:
            cur.wait_stream(st)
