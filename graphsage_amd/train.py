"""Seed-edge data-parallel training of the link model.

The reference trains through Keras: ``concat(user_emb, item_emb) -> Dense(1,
sigmoid)`` (``link_utils.py:167-168,244-247``), mean-squared-error loss and
``Adam(lr=1e-2)`` (``train_movielens.py:183-189``), one process.  Here the same
step runs one process per GPU: every rank holds the whole graph and feature
tables, takes its own shard of the seed edges (``flow(rank=, world_size=)``),
runs sampling + forward + backward through the HIP kernels, and the ranks
average ALL gradients with a single RCCL all-reduce over one flat fp32 bucket
(a few thousand to a few hundred thousand floats: latency-bound, so one
collective per step is the right granularity on xGMI).

On the GPU the link head with its loss and gradients (``gs_link_head_mse_fwd_bwd``)
and the optimizer update (``gs_adam_step`` over flat parameter storage) are HIP
kernels too; the torch formulations below them serve host tensors (tests under
gloo) and the autograd path.
"""
import math
import time

import numpy as np
import torch

__all__ = ['LinkHead', 'FlatGradBucket', 'Adam', 'Trainer']


class LinkHead:
    """concat -> Dense(1) -> sigmoid, Keras initialisers (glorot-uniform kernel,
    zero bias); output shape (B, 1) as after ``Reshape((1,))``."""

    def __init__(self, emb_dim, device, seed=None):
        gen = torch.Generator(device='cpu')
        if seed is None:
            gen.seed()
        else:
            gen.manual_seed(int(seed))
        limit = math.sqrt(6.0 / (2 * emb_dim + 1))
        k = (torch.rand(2 * emb_dim, 1, generator=gen, dtype=torch.float64) * 2 - 1) * limit
        self.kernel = k.to(torch.float32).to(device).requires_grad_(True)
        self.bias = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)

    def parameters(self):
        return [self.kernel, self.bias]

    def __call__(self, x_out):
        user_emb, item_emb = x_out
        z = torch.cat([user_emb, item_emb], dim=-1) @ self.kernel + self.bias
        return torch.sigmoid(z).reshape(-1, 1)


class FlatGradBucket:
    """All gradients in one contiguous fp32 buffer.

    ``p.grad`` of every parameter is a view into the buffer, so backward writes
    straight into it and the all-reduce needs no packing or unpacking.
    """

    def __init__(self, params):
        self.params = list(params)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        # parameters move into one flat buffer as well (their tensors become views of it), so the
        # optimizer update is a single elementwise pass
        self.flat_params = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            with torch.no_grad():
                self.flat_params[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_params[off:off + n].view_as(p)
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        # prepared kernel calls cache the weights' device addresses: tell them the weights moved
        from . import kernels
        kernels.parameters_moved()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        """Sum over ranks, divide by the world size (no-op for a single process)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return self.flat
        world = dist.get_world_size(group)
        if world > 1:
            self._sum_over_ranks(group)
            self.flat.div_(world)
        return self.flat

    def host_copy(self, slot=0):
        """A pinned host buffer of the bucket's size (two slots), for the gloo backend."""
        if getattr(self, '_host', None) is None:
            self._host = [torch.empty(self.flat.numel(), dtype=torch.float32).pin_memory() for _ in range(2)]
        return self._host[slot]

    def _sum_over_ranks(self, group):
        """RCCL ("nccl") collectives are ordered on the stream like kernels.  gloo is a host library: the bucket
        goes to the host, is summed there and comes back, each copy ordered with the device's work (gloo's own
        handling of device tensors, on worker threads with streams of their own, is not relied on)."""
        import torch.distributed as dist
        if self.flat.is_cuda and dist.get_backend(group) == 'gloo':
            host = self.host_copy()
            host.copy_(self.flat)                            # waits for the gradients
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            self.flat.copy_(host, non_blocking=True)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    def all_reduce_sum(self, group=None):
        """Sum over ranks, left undivided; returns the factor (1 / world) the
        optimizer applies to the gradients as it reads them, which saves the
        division pass.  1.0 for a single process."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        world = dist.get_world_size(group)
        if world <= 1:
            return 1.0
        self._sum_over_ranks(group)
        return 1.0 / world


class Adam:
    """Keras-default Adam on one flat parameter/gradient pair: beta1 0.9, beta2
    0.999, epsilon 1e-7, no amsgrad; ``lr_t = lr * sqrt(1-b2^t) / (1-b1^t)``,
    ``p -= lr_t * m / (sqrt(v) + eps)`` (the TF2 formulation)."""

    def __init__(self, params, grads_flat, lr=1e-2, beta_1=0.9, beta_2=0.999, epsilon=1e-7, params_flat=None):
        self.params = list(params)
        self.g = grads_flat
        self.p_flat = params_flat
        self.lr_dev = None
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.m = torch.zeros_like(grads_flat)
        self.v = torch.zeros_like(grads_flat)
        self.t = 0

    def next_rate(self):
        """Counts a step and returns its bias-corrected rate (for a caller that
        hands the rate to a replayed graph itself)."""
        self.t += 1
        return self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)

    @torch.no_grad()
    def step(self, launch=True, grad_scale=1.0):
        """launch=False only refreshes the device-side rate (the update kernel
        itself is then part of a replayed HIP graph).  grad_scale multiplies the
        gradients as they are read (1 / world after a summing all-reduce)."""
        lr_t = self.next_rate()
        if self.p_flat is not None and self.g.is_cuda:
            from . import kernels
            if not launch:                               # graph replay: the rate travels through memory
                self.lr_dev.fill_(lr_t)
                return
            # launched here: the rate goes by value, one launch and no fill
            kernels.adam_step(self.p_flat, self.g, self.m, self.v, lr_t, self.b1, self.b2, self.eps,
                              grad_scale=grad_scale)
            return
        if grad_scale != 1.0:
            self.g.mul_(grad_scale)
        self.m.mul_(self.b1).add_(self.g, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(self.g, self.g, value=1 - self.b2)
        upd = self.m / (self.v.sqrt() + self.eps)
        off = 0
        for p in self.params:
            p.add_(upd[off:off + p.numel()].view_as(p), alpha=-lr_t)
            off += p.numel()


class Trainer:
    """One optimisation step = batch -> embeddings -> head -> MSE -> backward ->
    gradient all-reduce -> Adam."""

    def __init__(self, model, head=None, lr=1e-2, group=None, engine='auto', graph=None, batch_size=None,
                 overlap='none', collective_at_world_one=False):
        """engine: 'autograd' runs the model through torch.autograd Functions (any
        shape); 'plan' uses the static launch plan of ``SageTrainEngine`` (needs
        ``graph`` and ``batch_size``); 'auto' picks the plan when it applies.

        overlap: how ``step_graph`` runs the gradient all-reduce when there is more than one rank.
        'none': replayed graph (sampling, forward, backward), then the collective, then one update launch.
        'capture': the collective and the update are captured INSIDE the graph, a step is one replay (needs a
        backend whose collectives are stream operations: RCCL).
        'delayed': step k's gradients are copied aside and all-reduced on a second stream while step k + 1's
        forward and backward run; they are applied after step k + 1's backward (the weights a step reads are one
        update behind; ``flush()`` applies the last one).
        collective_at_world_one: treat a process group of one rank like a larger one (the collective is issued
        all the same): lets a single GPU rehearse the 'capture' and 'delayed' paths."""
        if overlap not in ('none', 'capture', 'delayed'):
            raise ValueError("overlap must be 'none', 'capture' or 'delayed'")
        self.overlap = overlap
        self._collective_always = bool(collective_at_world_one)
        self.fuse_update = True                              # let the backward pass apply the optimizer step where it can
        self._pending = None                                 # delayed mode: (work, buffer, 1 / world) of the last step
        self._send = None
        self._coll_stream = None
        self._n_graph_steps = 0
        self._n_plan_steps = 0                               # steps taken through the static plan (launched directly or replayed)
        self.model = model.build()
        dev = self.model.parameters()[0].device
        emb_dim = self.model.dims[-1]['user']
        self.head = head or LinkHead(emb_dim, dev)
        self.params = self.model.parameters() + self.head.parameters()
        self.bucket = FlatGradBucket(self.params)
        self.opt = Adam(self.params, self.bucket.flat, lr=lr, params_flat=self.bucket.flat_params)
        self._head_step = None
        self._fwd_engine = None
        self._fwd_pipe = None                                # two batches in flight for evaluate / predict
        self._graph_obj, self._batch_size = graph, batch_size
        self.group = group
        self.engine = None
        self.engine_fallback_reason = None                   # why engine='auto' took the autograd route, if it did
        if not isinstance(self.head, LinkHead):
            # heads other than concat -> Dense(1, sigmoid) are torch modules: the model trains through autograd
            if engine == 'plan':
                raise ValueError("engine='plan' needs the concat -> Dense(1, sigmoid) head (train.LinkHead)")
            engine = 'autograd'
        if engine in ('auto', 'plan') and graph is not None and batch_size is not None:
            from .train_engine import SageTrainEngine
            from ._lib import GsageError
            try:
                self.engine = SageTrainEngine(graph, self.model, batch_size)
            except (GsageError, ValueError) as ex:          # a shape the plan does not cover; anything else is a bug
                if engine == 'plan':
                    raise
                self.engine_fallback_reason = str(ex)
        elif engine == 'plan':
            raise ValueError("engine='plan' needs graph and batch_size")
        self.broadcast_parameters()

    def broadcast_parameters(self):
        """Rank 0's initial weights everywhere (ranks seed their own generators)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            with torch.no_grad():
                flat = torch.cat([p.reshape(-1) for p in self.params])
                if flat.is_cuda and dist.get_backend(self.group) == 'gloo':
                    host = flat.cpu()                        # gloo is a host library: hand it host memory
                    dist.broadcast(host, src=0, group=self.group)
                    flat = host.to(flat.device)
                else:
                    dist.broadcast(flat, src=0, group=self.group)
                off = 0
                for p in self.params:
                    p.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()

    def loss(self, batch, targets):
        pred = self.head(self.model(batch))
        y = torch.as_tensor(targets, dtype=torch.float32, device=pred.device).reshape(pred.shape)
        return ((pred - y) ** 2).mean(), pred

    def step_device(self, seeds, targets, walk_ids=None, rng_seed=0):
        """Training step on device-resident inputs through the static plan:
        ``seeds`` int32[2B] roots (sources then destinations), ``targets``
        float32[B] on the GPU."""
        from . import kernels
        eng = self.engine
        self._prepare_head_step()
        self._n_plan_steps += 1
        # every gradient of the bucket is overwritten by this step's kernels: no zeroing pass
        emb_u, emb_i = eng.forward_train(seeds, walk_ids, rng_seed)
        if not self._ranks()[0] and self._update_in_backward():
            # one process: nothing stands between the gradients and the update, the backward pass applies it
            loss, pred = self._head_and_backward(emb_u, emb_i, targets, self._adam_args(lr=self.opt.next_rate()))
        else:
            loss, pred = self._head_and_backward(emb_u, emb_i, targets)
            self.opt.step(grad_scale=self.bucket.all_reduce_sum(self.group))
        return loss, pred.reshape(-1, 1)

    def step_is_short(self):
        """True when a step on this process is the handful of launches of the two-hop one-kernel route (forward,
        backward, reduction, at most an update and the copies of a padded model): launched one by one they leave
        no gap the host could not fill, while a replayed graph pays its launch (8 us between two replays on
        MI355X) and a staging kernel for its inputs.  Measured, direct against replayed: 158 / 166 us at the
        benchmark shape, 88 / 96 us at the MovieLens shape, 232 / 233 us with 64 hidden units; the layered routes
        (dozens of launches on two streams) are faster replayed (three hops: 1.46 against 1.49 ms)."""
        return self.engine is not None and not self._ranks()[0] and bool(getattr(self.engine, 'walk_train', False))

    def step_auto(self, seeds, targets, walk_ids, rng_seed=0):
        """``step_device`` where the step is short (``step_is_short``), ``step_graph`` otherwise."""
        if self.step_is_short():
            return self.step_device(seeds, targets, walk_ids, rng_seed)
        return self.step_graph(seeds, targets, walk_ids, rng_seed)

    def _update_in_backward(self):
        eng = self.engine
        return bool(self.fuse_update and eng is not None and (eng.tail_kernel or getattr(eng, 'fused_backward3', False))
                    and self.opt.p_flat is not None
                    and eng.can_update_in_backward(self.bucket.flat, (self.head.kernel.grad, self.head.bias.grad)))

    def _adam_args(self, lr=None, lr_dev=None):
        o = self.opt
        d = {'grads': self.bucket.flat, 'params': o.p_flat, 'm': o.m, 'v': o.v, 'b1': o.b1, 'b2': o.b2, 'eps': o.eps}
        if lr_dev is not None:
            d['lr_dev'] = lr_dev
        else:
            d['lr'] = lr
        return d

    def _head_and_backward(self, emb_u, emb_i, targets, adam=None):
        """Head + loss + the engine's backward pass; on the two-hop route the head is evaluated inside
        the first kernel of the backward pass instead of two launches of its own.  ``adam``: the optimizer
        step as well (``_adam_args``; only when ``_update_in_backward()``)."""
        eng, hs = self.engine, self._head_step
        if adam is not None:
            eng.backward_head(self.head.kernel, self.head.bias, targets, self.head.kernel.grad, self.head.bias.grad,
                              hs.pred, hs.loss, adam=adam)
            return hs.loss, hs.pred
        if eng.tail_kernel or getattr(eng, 'fused_backward3', False):
            if not (isinstance(targets, torch.Tensor) and targets.is_cuda and targets.dtype == torch.float32
                    and targets.is_contiguous() and targets.numel() == eng.batch):
                raise ValueError("targets must be a contiguous float32 GPU tensor of %d entries" % eng.batch)
            eng.backward_head(self.head.kernel, self.head.bias, targets, self.head.kernel.grad, self.head.bias.grad,
                              hs.pred, hs.loss)
            return hs.loss, hs.pred
        loss, pred = hs.run(emb_u, emb_i, targets)
        eng.backward(hs.g_user, hs.g_item)
        return loss, pred

    def _prepare_head_step(self):
        """(Re)builds the prepared head call and drops the engine's prepared weight-gradient calls when the
        parameters have been moved since they were made (another Trainer built on the same model).
        Returns True when it did."""
        from . import kernels
        if self._head_step is not None and self._head_step.epoch == kernels.storage_epoch():
            return False
        eng = self.engine
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() < self.bucket.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.bucket.flat.data_ptr() + self.bucket.flat.numel() * 4:
                raise kernels._lib.GsageError("this Trainer's parameters now live in another Trainer's bucket")
        self._head_step = kernels.LinkHeadStep(eng.batch, self.head.kernel.shape[0] // 2, self.head.kernel,
                                               self.head.bias, self.head.kernel.grad, self.head.bias.grad,
                                               self.head.kernel.device)
        eng._wgrad_pairs = None
        eng._wgrad_l0_pair = None
        eng._bwd3 = None
        eng._bwd_all = None
        return True

    # -- HIP-graph replay of the step ---------------------------------------------------
    def _ranks(self):
        """(more than one rank -- or one rank asked to behave like several --, world size)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False, 1
        world = dist.get_world_size(self.group)
        return world > 1 or self._collective_always, world

    def _graph_body(self, with_update, with_collective=False):
        import torch.distributed as dist
        from . import kernels
        eng = self.engine
        emb_u, emb_i = eng.forward_train(eng.seeds_buf, self._walk_buf, self._graph_seed)
        if with_update and not with_collective and self._update_in_backward():
            self._head_and_backward(emb_u, emb_i, self._target_buf, self._adam_args(lr_dev=self.opt.lr_dev))
            return
        self._head_and_backward(emb_u, emb_i, self._target_buf)
        scale = 1.0
        if with_collective:
            # the sum over ranks as a node of the graph; the update reads the gradients times 1 / world
            dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
            scale = 1.0 / dist.get_world_size(self.group)
        if with_update:
            kernels.adam_step(self.opt.p_flat, self.opt.g, self.opt.m, self.opt.v, 0.0, self.opt.b1, self.opt.b2,
                              self.opt.eps, grad_scale=scale, lr_t_device=self.opt.lr_dev)

    def _start_delayed_all_reduce(self):
        """Copies this step's gradients aside and starts their all-reduce on the collective stream; returns what
        ``_apply_delayed`` needs.  The next replay may overwrite the bucket at once."""
        import torch.distributed as dist
        main = torch.cuda.current_stream()
        scale = 1.0 / dist.get_world_size(self.group)
        if self._send is None:
            self._send = [torch.empty_like(self.bucket.flat) for _ in range(2)]
            self._coll_stream = torch.cuda.Stream()
        buf = self._send[self._n_graph_steps % 2]
        if dist.get_backend(self.group) == 'gloo':
            # a host library: the gradients travel as a pinned host copy, summed on gloo's threads
            host = self.bucket.host_copy(self._n_graph_steps % 2)
            host.copy_(self.bucket.flat)
            work = dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return work, buf, scale, host
        buf.copy_(self.bucket.flat)
        side = self._coll_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work, buf, scale, None

    def _apply_delayed(self, pending):
        from . import kernels
        work, buf, scale, host = pending
        if host is not None:
            work.wait()
            buf.copy_(host, non_blocking=True)
        else:
            with torch.cuda.stream(self._coll_stream):
                work.wait()
            torch.cuda.current_stream().wait_stream(self._coll_stream)
        kernels.adam_step(self.opt.p_flat, buf, self.opt.m, self.opt.v, self.opt.next_rate(), self.opt.b1, self.opt.b2,
                          self.opt.eps, grad_scale=scale)

    def flush(self):
        """Applies the update a 'delayed' step left pending (call before reading or saving the weights)."""
        if self._pending is not None:
            pending, self._pending = self._pending, None
            self._apply_delayed(pending)

    def step_graph(self, seeds, targets, walk_ids, rng_seed=0):
        """``step_device`` as a captured HIP graph: the ~30 launches of a step are
        replayed by one call, which removes the per-launch host cost.  Inputs are
        copied into static buffers; the sampling key is fixed at capture (draws
        still differ per link through ``walk_ids``), so a new ``rng_seed`` - a new
        epoch - recaptures.  With more than one rank the all-reduce runs as
        ``overlap`` says (constructor): after the graph, inside it, or beside the
        next step."""
        from . import kernels
        eng = self.engine
        multi, _ = self._ranks()
        in_graph = multi and self.overlap == 'capture'       # collective + update are graph nodes
        outside = multi and not in_graph                     # the graph ends after backward
        if self._prepare_head_step():
            self._graph = None                               # captured addresses are those of the old layout
        if getattr(self, '_graph', None) is None or self._graph_seed != rng_seed:
            dev = eng.seeds_buf.device
            self._walk_buf = torch.zeros(2 * eng.batch, dtype=torch.int64, device=dev)
            self._target_buf = torch.zeros(eng.batch, dtype=torch.float32, device=dev)
            self.opt.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
            self._graph_seed = rng_seed
            eng.seeds_buf.copy_(seeds)
            self._walk_buf.copy_(walk_ids)
            self._target_buf.copy_(targets)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                # warm-up outside capture (lazy plan construction)
                self.opt.lr_dev.zero_()                  # rate 0: the warm-up must not move the weights
                m_save, v_save = self.opt.m.clone(), self.opt.v.clone()
                self._graph_body(not outside, in_graph)  # (an in-graph collective runs here too, on every rank)
                self.opt.m.copy_(m_save)
                self.opt.v.copy_(v_save)
            torch.cuda.current_stream().wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            # thread-local capture checks: a communicator's watchdog thread may poll its own events
            # while this thread captures
            torch.cuda.synchronize()
            with torch.cuda.graph(self._graph, capture_error_mode='thread_local'):
                self._graph_body(not outside, in_graph)
        def fits(src, dst):
            return (isinstance(src, torch.Tensor) and src.device == dst.device and src.dtype == dst.dtype
                    and src.is_contiguous() and src.numel() == dst.numel())
        if fits(seeds, eng.seeds_buf) and fits(walk_ids, self._walk_buf) and fits(targets, self._target_buf):
            # the usual case (rows of an epoch staged on the device): inputs and rate in one launch
            kernels.stage_batch(seeds, eng.seeds_buf, walk_ids, self._walk_buf, targets, self._target_buf,
                                0.0 if outside else self.opt.next_rate(), None if outside else self.opt.lr_dev)
        else:
            eng.seeds_buf.copy_(seeds)
            self._walk_buf.copy_(walk_ids)
            self._target_buf.copy_(targets)
            if not outside:
                self.opt.step(launch=False)              # sets the device-side rate for this step
        self._graph.replay()
        if outside and self.overlap == 'delayed':
            # this step's gradients travel while the next step computes; the previous step's arrive now
            started = self._start_delayed_all_reduce()
            if self._pending is not None:
                self._apply_delayed(self._pending)
            self._pending = started
        elif outside:
            # sampling + forward + backward replayed as one graph; then ONE collective (sum) and ONE update
            # launch that applies 1 / world to the gradients as it reads them
            self.opt.step(grad_scale=self.bucket.all_reduce_sum(self.group))
        self._n_graph_steps += 1
        self._n_plan_steps += 1
        return self._head_step.loss, self._head_step.pred.reshape(-1, 1)

    # -- Keras-style loops over a LinkSequence ------------------------------------------
    def _metrics(self):
        from .metrics import AUC, BinaryAccuracy, RootMeanSquaredError
        return [RootMeanSquaredError(), BinaryAccuracy(), AUC()]

    def _common_batches(self, seq):
        """Number of batches every rank can take (a collective per step needs all
        ranks to make the same number of steps)."""
        import torch.distributed as dist
        n = len(seq)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            t = torch.tensor([n], dtype=torch.int64, device=self.bucket.flat.device)
            if t.is_cuda and dist.get_backend(self.group) == 'gloo':
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            n = int(t.item())
        return n

    def fit(self, train_seq, validation_data=None, epochs=1, verbose=1):
        """The loop ``Model.fit(train_gen, validation_data=test_gen, epochs=...)``
        runs in the reference (``train_movielens.py:200-207``): one optimisation
        step per batch, the compiled metrics (``:187-189``) accumulated over the
        epoch, validation after it, ``on_epoch_end`` of the sequence.  Full
        batches go through the static plan when there is one, short ones through
        autograd.  Returns {'loss': [...], 'auc': [...], 'val_loss': [...], ...}."""
        history = {}
        dev = self.bucket.flat.device
        for epoch in range(epochs):
            t_epoch = time.perf_counter()
            mets = self._metrics()
            ys, preds, losses = [], [], []                     # metrics are sums: taken once, after the pass
            staged = None
            if (self.engine is not None and getattr(train_seq, '_indexed', False) and train_seq.targets is not None
                    and train_seq.batch_size == self.engine.batch):
                staged = train_seq.epoch_on_device()           # every full batch of the epoch, one copy
            for i in range(self._common_batches(train_seq)):
                if staged is not None and i < staged['n_batches']:
                    y_dev = staged['targets'][i]
                    # replayed HIP graph; loss and pred are its static outputs, consumed (in stream
                    # order) by the metric updates below before the next replay overwrites them
                    loss, pred = self.step_auto(staged['roots'][i], y_dev, staged['walk_ids'][i], staged['key'])
                else:
                    batch, y = train_seq[i]
                    y_dev = torch.as_tensor(np.asarray(y, dtype=np.float32), device=dev)
                    loss, pred = self.step(batch, y_dev)
                ys.append(y_dev.reshape(-1))
                preds.append(pred.detach().reshape(-1).clone())
                losses.append(loss.detach().reshape(1).clone())
            self.flush()                                       # (delayed mode: the last step's update)
            logs = self._epoch_logs(mets, ys, preds, losses)
            self._raise_sampling_errors()
            if validation_data is not None:
                logs.update({'val_' + k: v for k, v in self.evaluate(validation_data).items()})
            for k, v in logs.items():
                history.setdefault(k, []).append(v)
            if verbose:
                print("Epoch %d/%d - %.1fs - %s" % (epoch + 1, epochs, time.perf_counter() - t_epoch,
                                                    " - ".join("%s: %.4f" % kv for kv in logs.items())))
            train_seq.on_epoch_end()
        return history

    @staticmethod
    def _epoch_logs(mets, ys, preds, losses, chunk=1 << 16):
        """{'loss', metric names...} of one pass from its per-batch targets, scores
        and (for training, where the weights move between batches) batch losses;
        without ``losses`` the loss is the mean squared error of the scores."""
        if not ys:
            return dict({'loss': 0.0}, **{m.name: m.result() for m in mets})
        sizes = [t.numel() for t in ys]
        y_all, p_all = torch.cat(ys), torch.cat(preds)
        if losses is None:
            loss = float(((p_all.double() - y_all.double()) ** 2).mean())
        else:
            w = torch.tensor(sizes, dtype=torch.float64, device=y_all.device)
            loss = float((torch.cat(losses).double() * w).sum() / w.sum())
        for lo in range(0, y_all.numel(), chunk):
            for m in mets:
                m.update_state(y_all[lo:lo + chunk], p_all[lo:lo + chunk])
        return dict({'loss': loss}, **{m.name: m.result() for m in mets})

    def _raise_sampling_errors(self):
        """The engines only flag a node without neighbors or an unknown root (the
        reference raises for either while it builds the batch); surface the
        flags as the same exceptions, once per pass over a sequence."""
        from .sampler import BreadthFirstWalker
        for eng in (self.engine, self._fwd_engine):
            if eng is not None:
                BreadthFirstWalker.raise_on_error(eng.err_flag)

    @torch.no_grad()
    def _scores(self, seq, i):
        """(scores (b, 1), targets) of batch i: through the forward engine when the
        batch is a full one of a keyed sequence, else through the model."""
        if (self._graph_obj is not None and self._batch_size and getattr(seq, '_indexed', False)
                and seq.batch_size == int(self._batch_size) and not (self.model.dropout and self.model.training)):
            staged = seq.epoch_on_device()
            if i < staged['n_batches']:
                if self._fwd_engine is None:
                    from .engine import SageForwardEngine
                    self._fwd_engine = SageForwardEngine(self._graph_obj, self.model, self._batch_size)
                emb = self._fwd_engine.forward(staged['roots'][i], staged['walk_ids'][i], rng_seed=staged['key'])
                y = None if staged['targets'] is None else staged['targets'][i]
                return self.head(emb), y
        batch, y = seq[i]
        return self.head(self.model(batch)), y

    @torch.no_grad()
    def _score_stream(self, seq):
        """(scores, targets) of every batch of ``seq`` in order.  Full batches of a keyed sequence go through the
        forward engine with two batches in flight where its route allows (``SageForwardEngine.pipeline``): batch
        i + 1 is launched before batch i's embeddings are scored, and a slot's rows are scored on this stream
        before the slot's stream is allowed to write them again (``submit`` makes it wait for this stream)."""
        n = len(seq)
        use_engine = (self._graph_obj is not None and self._batch_size and getattr(seq, '_indexed', False)
                      and seq.batch_size == int(self._batch_size) and not (self.model.dropout and self.model.training))
        n_eng = 0
        if use_engine:
            staged = seq.epoch_on_device()
            n_eng = min(n, staged['n_batches'])
            if self._fwd_engine is None:
                from .engine import SageForwardEngine
                self._fwd_engine = SageForwardEngine(self._graph_obj, self.model, self._batch_size)
            eng = self._fwd_engine
        if n_eng >= 2 and (eng.walk_kernel or eng.subtree_kernel):
            if self._fwd_pipe is None:
                self._fwd_pipe = eng.pipeline(2)
            pipe, cur = self._fwd_pipe, torch.cuda.current_stream()

            def score(i, out):
                cur.wait_event(out[2])
                y = None if staged['targets'] is None else staged['targets'][i]
                return self.head((out[0], out[1])), y
            pending = None
            for i in range(n_eng):
                out = pipe.submit(staged['roots'][i], staged['walk_ids'][i], rng_seed=staged['key'])
                if pending is not None:
                    yield score(i - 1, pending)
                pending = out
            yield score(n_eng - 1, pending)
            pipe.drain()
        else:
            for i in range(n_eng):
                yield self._scores(seq, i)
        for i in range(n_eng, n):
            yield self._scores(seq, i)

    def _inference(self):
        """Context in which the model runs with dropout off, as Keras runs ``evaluate``, ``predict`` and the
        validation pass of ``fit`` (``train_movielens.py:200-207``); the training flag is restored after."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            was = self.model.training
            self.model.eval()
            try:
                yield
            finally:
                self.model.train(was)
        return ctx()

    @torch.no_grad()
    def evaluate(self, seq):
        """Loss and metrics of the current weights over a sequence (no update, no dropout)."""
        dev = self.bucket.flat.device
        ys, preds = [], []
        self.flush()
        with self._inference():
            for pred, y in self._score_stream(seq):
                ys.append((y if isinstance(y, torch.Tensor)
                           else torch.as_tensor(np.asarray(y, dtype=np.float32), device=dev)).reshape(-1))
                preds.append(pred.reshape(-1))
        out = self._epoch_logs(self._metrics(), ys, preds, None)
        self._raise_sampling_errors()
        return out

    @torch.no_grad()
    def predict(self, seq):
        """Scores of every link of a sequence, in sequence order: float32 [n, 1] (no dropout)."""
        self.flush()
        with self._inference():
            return torch.cat([pred for pred, _ in self._score_stream(seq)])

    def step(self, batch, targets):
        self.bucket.zero()
        loss, pred = self.loss(batch, targets)
        loss.backward()
        self.bucket.all_reduce_mean(self.group)
        self.opt.step()
        return loss.detach(), pred.detach()
